#!/usr/bin/env python3
"""Per-phase timing of the fused GAT-view BACKWARD kernel (csrc/dvgr_gatb.hip) from the phase-clock debug build
(make -C dualvgr-videoqa_amd/csrc clk).  usage: DVGR_LIB=dualvgr-videoqa_amd/libdvgr_hip_clk.so python tools/gatb_clocks.py [B=64] [num_clips=20]
Phases: K loop (incl. prologue) | B0 loads issued, acc -> LDS | B1 alpha, dP, Wh | B2 dalpha | B3 softmax bwd | B4 dv, dZ | B5 dWh, ds, da"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L_, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
Lr, T, D, WD = 2, 15, 768, 300
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3").cuda().train()
m.single_stream = True
inp = synth.unit_inputs(B, N, T, D, WD, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
for k in ("A", "M"):
    t[k].requires_grad_(True)
lib = L_.load()
lib.dvgr_dbg_gatb_clocks.argtypes = [C.c_void_p, C.c_int]
for it in range(5):
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    torch.autograd.backward([raw[0], *raw[3:]], [torch.ones_like(raw[0])] + [torch.ones_like(r) for r in raw[3:]])
    torch.cuda.synchronize()
n = 1024 * 8
buf = np.zeros(n, dtype=np.int64)
assert lib.dvgr_dbg_gatb_clocks(buf.ctypes.data, n) == 0
c = buf.reshape(1024, 8).astype(np.float64) / 100.0       # 100 MHz -> us
c = c[c[:, 7] > 0]
d = np.diff(c, axis=1)
names = ["K loop", "B0", "B1", "B2", "B3", "B4", "B5"]
print(f"B={B} N={N}: {len(c)} workgroups; phase durations us (mean / max):")
for i, nm in enumerate(names):
    print(f"  {nm:7s} {d[:, i].mean():7.2f} {d[:, i].max():7.2f}")
print("per-WG total mean", round((c[:, 7] - c[:, 0]).mean(), 2), "kernel span", round(c[:, 7].max() - c[:, 0].min(), 2),
      "start spread", round(c[:, 0].max() - c[:, 0].min(), 2))
