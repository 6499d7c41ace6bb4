#!/usr/bin/env python3
"""The forward -> backward valley of the C2 step WITHOUT a profiler (rocprofv3 slows the host's launch path 2-3 x, and the call
boundary between dvgr_unit_forward and dvgr_unit_backward then shows as a 30 us gap that a free-running step does not have).
Two hipEvents of the library's own probe machinery: one recorded in front of the S6 launch (= right behind the last
k_sf_combine_fwd), one in front of the last layer's dz contraction (= right behind the first k_sf_bwd); their distance minus
k_sf_bwd's own duration (rocprofv3: profiles/r04_c2_summary.txt) is the valley.   GPU only.   python tools/valley_probe.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L_, synth

B, N, T, D, WD, Lr = 64, 20, 15, 768, 300, 3
torch.manual_seed(9)
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")     # (its own random init)
m = m.cuda().train(True)
m.loss_in_forward = (1.0, 1e-8, 1.0)
m.reuse_buffers = True
inp = synth.unit_inputs(B, N, T, D, WD, seed=9)
t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
for k in ("A", "M", "dq", "words"):
    t[k].requires_grad_(True)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]


def step(i=None):
    m.zero_grad_flat()
    for k in ("A", "M", "dq", "words"):
        t[k].grad = None
    if i is not None and i % 4 == 0:
        m.set_probe(L_.PROBE_FWD_MFB_GEMM, 0, ev[i][0], ev[i][1])
    else:
        m.set_probe(0)
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    _, g = m.last_loss
    if i is not None and i % 4 == 0:
        m.set_probe(L_.PROBE_BWD_SF_DX, Lr - 1, ev[i][2], ev[i][3])
    torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])


for _ in range(30):
    step()
torch.cuda.synchronize()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
d = np.array([ev[i][0].elapsed_time(ev[i][2]) * 1e3 for i in range(0, steps, 4)])
s6 = np.array([ev[i][0].elapsed_time(ev[i][1]) * 1e3 for i in range(0, steps, 4)])
print(f"front of S6 -> behind the first k_sf_bwd: median {np.median(d):.1f} us (min {d.min():.1f}, max {d.max():.1f}) over {len(d)} probed steps")
print(f"  of which the S6 launch itself (k_mfb_fwd): {np.median(s6):.1f} us")
print("  subtract k_sf_bwd's own duration (profiles/r04_c2_summary.txt) for the distance last k_sf_combine_fwd -> first k_sf_bwd")
