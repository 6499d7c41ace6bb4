#!/usr/bin/env python3
"""Host enqueue time vs device time of one bench.py training step (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L_, synth

B, N, Lr, T, D, WD = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 20, (int(sys.argv[2]) if len(sys.argv) > 2 else 3), 15, 768, 300
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3").cuda().train()
m.loss_in_forward = (1.0, 1e-8, 1.0)
m.reuse_buffers = True
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

for _ in range(20):
    step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step; device-complete {1e3 * (t2 - t0) / n:.3f} ms/step; "
      f"host finished {1e3 * (t2 - t1):.2f} ms before the device")
# forward-only and backward-only host cost
t0 = time.perf_counter()
raws = []
for _ in range(n):
    raws.append(m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"]))
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"forward host enqueue {1e3 * (t1 - t0) / n:.3f} ms/call")
