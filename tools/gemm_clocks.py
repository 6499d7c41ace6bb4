#!/usr/bin/env python3
"""Where a workgroup of gemm_planes_kernel spends its time (phase-clock debug build, `make clk`).
usage: DVGR_LIB=dualvgr-videoqa_amd/libdvgr_hip_clk.so python tools/gemm_clocks.py [mode M N K]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dualvgr_videoqa_amd import _lib as L

lib = L.load()
lib.dvgr_dbg_gemm_clocks.argtypes = [C.c_void_p, C.c_int]
mode, M, N, K = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (0, 2560, 1536, 768)
st = lambda: torch.cuda.current_stream().cuda_stream


def planes(x):
    n = x.numel()
    pl = torch.empty(2 * n, device="cuda", dtype=torch.bfloat16)
    assert lib.dvgr_dbg_split(x.data_ptr(), pl.data_ptr(), n, n, st()) == 0
    return pl, n


shapes = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[mode]
A, B = torch.randn(*shapes[0], device="cuda"), torch.randn(*shapes[1], device="cuda")
pa, na = planes(A)
pb, nb = planes(B)
Cm = torch.empty(M, N, device="cuda")
def call():
    rc = lib.dvgr_dbg_gemm_planes(mode, M, N, K, A.data_ptr(), pa.data_ptr(), A.stride(0), na, B.data_ptr(), pb.data_ptr(),
                                  B.stride(0), nb, Cm.data_ptr(), Cm.stride(0), 0, 0, 2, st())
    assert rc == 0, rc
for _ in range(5):
    call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
tiles = ((M + 127) // 128) * ((N + 127) // 128)
buf = np.zeros(tiles * 8, dtype=np.int64)
assert lib.dvgr_dbg_gemm_clocks(buf.ctypes.data, tiles * 8) == 0
c = buf.reshape(tiles, 8).astype(np.float64) / 100.0
names = ["cons: barrier wait", "cons: reads+mfma", "cons: epilogue", "-", "load: dma wait", "load: barrier wait", "load: issue", "-"]
print(f"mode {mode} M{M} N{N} K{K}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, {tiles} tiles")
for k, nme in enumerate(names):
    if nme != "-":
        print(f"   {nme:22s} mean {c[:, k].mean():7.2f} us   max {c[:, k].max():7.2f}")
