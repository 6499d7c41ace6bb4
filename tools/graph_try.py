#!/usr/bin/env python3
"""Feasibility probe: capture one eval-mode training step (forward + loss + backward) of the unit into a hipGraph
through torch.cuda.graph and replay it; compares eager and replayed step times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import synth

B, N, Lr, T, D, WD = [int(x) for x in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 20, 3, 15, 768, 300))]
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3").cuda().eval()
m.loss_in_forward = (1.0, 1e-8, 1.0)
inp = synth.unit_inputs(B, N, T, D, WD, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
for k in ("A", "M", "dq", "words"):
    t[k].requires_grad_(True)

def step():
    m.zero_grad_flat()
    for k in ("A", "M", "dq", "words"):
        t[k].grad = None
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    _, g = m.last_loss
    torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])

def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print(f"eager: {timeit(step):.3f} ms/step")
ref = m.last_flat_grad.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step()
except Exception as e:          # noqa: BLE001
    print("capture failed:", repr(e)[:400])
    sys.exit(0)
torch.cuda.synchronize()
print(f"graph replay: {timeit(g.replay):.3f} ms/step")
print("grad max abs diff vs eager:", float((m.last_flat_grad - ref).abs().max()), "ref max", float(ref.abs().max()))
