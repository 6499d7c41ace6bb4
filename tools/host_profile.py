#!/usr/bin/env python3
"""Where the host time of a training step goes: C-ABI calls vs Python / autograd around them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L_, synth

B, N, Lr, T, D, WD = 64, 20, 3, 15, 768, 300
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3").cuda().train()
m.loss_in_forward = (1.0, 1e-8, 1.0)
m.reuse_buffers = True
inp = synth.unit_inputs(B, N, T, D, WD, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
for k in ("A", "M", "dq", "words"):
    t[k].requires_grad_(True)
lib = L_.load()
acc = {"fwd_c": 0.0, "bwd_c": 0.0}
f0, b0 = lib.dvgr_unit_forward, lib.dvgr_unit_backward
class Timed:
    def __init__(self, fn, key): self.fn, self.key = fn, key
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); acc[self.key] += time.perf_counter() - t0; return r
lib.dvgr_unit_forward = Timed(f0, "fwd_c")
lib.dvgr_unit_backward = Timed(b0, "bwd_c")

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
acc["fwd_c"] = acc["bwd_c"] = 0.0
n = 200
tf = tb = 0.0
t0 = time.perf_counter()
for _ in range(n):
    m.zero_grad_flat()
    for k in ("A", "M", "dq", "words"):
        t[k].grad = None
    a = time.perf_counter()
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    b = time.perf_counter()
    _, g = m.last_loss
    torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])
    c = time.perf_counter()
    tf += b - a; tb += c - b
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host per step {1e3 * (t1 - t0) / n:.3f} ms: forward_raw {1e3 * tf / n:.3f} (C call {1e3 * acc['fwd_c'] / n:.3f}), "
      f"backward {1e3 * tb / n:.3f} (C call {1e3 * acc['bwd_c'] / n:.3f})")
