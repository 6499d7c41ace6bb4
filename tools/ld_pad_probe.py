#!/usr/bin/env python3
"""Does the leading dimension of the operand planes matter to the contraction engine's K loop?  A row of a (rows, 768) bf16
plane is 1536 B = 12 lines of 128 B: consecutive rows start 12 lines apart, so if the L2 picks its channel from low line-address
bits, the 16-row pieces of one LDS-DMA instruction fall on a quarter of the channels.  This times the S4 / S2 / dz / dX / dW
launch shapes of a C2 step with the operands' rows padded by 0 ... 256 elements (the kernels take any ld % 8 == 0).  GPU only.
usage: python tools/ld_pad_probe.py [reps=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L

lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, N, D = 64, 20, 768
BN = B * N


def planes(x):
    n = x.numel()
    pl = torch.empty(2 * n, device="cuda", dtype=torch.bfloat16)
    assert lib.dvgr_dbg_split(x.data_ptr(), pl.data_ptr(), n, n, st()) == 0
    return pl, n


def run(name, mode, M, Nn, K, count, tag, pad_a, pad_b, pad_c=0, act=0, acc=0, atomic=0):
    shp = {0: ((M, K), (Nn, K)), 1: ((M, K), (K, Nn)), 2: ((K, M), (K, Nn))}[mode]
    Af = torch.randn(shp[0][0], shp[0][1] + pad_a, device="cuda")
    Bf = torch.randn(shp[1][0], shp[1][1] + pad_b, device="cuda")
    pa, na = planes(Af)
    pb, nb = planes(Bf)
    Cm = torch.zeros(count * M, Nn + pad_c, device="cuda")

    def call():
        rc = lib.dvgr_dbg_gemm_planes_batch(mode, M, Nn, K, 0, Af.data_ptr(), pa.data_ptr(), Af.stride(0), na, Bf.data_ptr(),
                                            pb.data_ptr(), Bf.stride(0), nb, Cm.data_ptr(), Cm.stride(0), act, acc, atomic, 2, count, tag,
                                            None, 0, st())
        assert rc == 0, rc
    for _ in range(10):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gf = 2.0 * M * Nn * K * count / 1e9
    print(f"{name:34s} pad A {pad_a:4d} B {pad_b:4d} C {pad_c:4d}: {us:7.2f} us  {gf / us * 1e3:7.1f} TFLOP/s", flush=True)
    return us


if __name__ == "__main__":
    for pads in [(0, 0, 0), (64, 64, 0), (32, 32, 0), (128, 128, 0), (64, 0, 0), (0, 64, 0), (64, 64, 64), (192, 192, 0), (256, 256, 0), (0, 0, 0)]:
        run("S4 fwd NT 4 x (1280 x 768 x 768)", 0, BN, D, D, 4, 2, *pads, act=1)
    for pads in [(0, 0, 0), (64, 64, 0), (64, 64, 64), (0, 0, 0)]:
        run("dz  bwd NN 2 x (2560 x 768 x 768)", 1, 2 * BN, D, D, 2, 6, *pads, acc=1)
        run("dW  bwd TN 6 x (768 x 768 x 1280)", 2, 768, D, BN, 6, 3, *pads, atomic=1)
