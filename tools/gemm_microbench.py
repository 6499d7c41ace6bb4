#!/usr/bin/env python3
"""Micro-benchmark of the contraction engine through the debug hook: time vs K gives the per-K-tile cost
(slope) and the fixed prologue/epilogue/launch cost (intercept).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L

lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream


def run(mode, M, N, K, prec, reps=30):
    if mode == 0:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    elif mode == 1:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    def call():
        rc = lib.dvgr_dbg_gemm(mode, M, N, K, 0, A.data_ptr(), None, A.stride(0), B.data_ptr(), None, B.stride(0),
                               C.data_ptr(), N, None, 0, 0, 0, 0, 0, prec, st())
        assert rc == 0
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


if __name__ == "__main__":
    shapes = [(1280, 1536), (2560, 768), (1536, 768), (768, 768)]
    for mode in (0, 1, 2):
        for (M, N) in shapes:
            for prec in (0, 2, 1):
                row = []
                for K in (32, 128, 256, 768, 1536, 2560):
                    row.append(run(mode, M, N, K, prec))
                fl = 2.0 * M * N * 768 / (row[3] * 1e-6) / 1e12
                print(f"mode {mode} M{M} N{N} prec {prec}: " + " ".join(f"K{k}:{v:6.1f}us" for k, v in zip((32, 128, 256, 768, 1536, 2560), row)) + f"  | K768: {fl:6.1f} TF")
