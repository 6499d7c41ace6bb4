#!/usr/bin/env python3
"""Per-phase timing of the attention kernels from the phase-clock debug build (make -C dualvgr-videoqa_amd/csrc clk).
usage: DVGR_LIB=dualvgr-videoqa_amd/libdvgr_hip_clk.so python tools/phase_clocks.py [fwd|bwd] [num_clips=20]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L_, synth

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, N, Lr, T, D, WD = 64, (int(sys.argv[2]) if len(sys.argv) > 2 else 20), 1, 20, 768, 300
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3").cuda()
m.single_stream = True
inp = synth.unit_inputs(B, N, T, D, WD, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
for k in ("A", "M"):
    t[k].requires_grad_(True)
lib = L_.load()
lib.dvgr_dbg_clocks.argtypes = [C.c_void_p, C.c_int]
for it in range(3):
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    if which == "bwd":
        torch.autograd.backward([raw[0], *raw[3:]], [torch.ones_like(raw[0])] + [torch.ones_like(r) for r in raw[3:]])
    torch.cuda.synchronize()
n = 1024 * 8
buf = np.zeros(n, dtype=np.int64)
assert lib.dvgr_dbg_clocks(buf.ctypes.data, n) == 0
c = buf.reshape(1024, 8).astype(np.float64) / 100.0       # 100 MHz -> us
k = 6 if which == "fwd" else 7
d = np.diff(c[:, :k], axis=1)
print(which, "phase durations us  mean:", np.round(d.mean(0), 2), " max:", np.round(d.max(0), 2))
print("per-WG total mean", round((c[:, k - 1] - c[:, 0]).mean(), 2), "kernel span", round(c[:, k - 1].max() - c[:, 0].min(), 2),
      "start spread", round(c[:, 0].max() - c[:, 0].min(), 2))
