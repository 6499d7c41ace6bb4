#!/usr/bin/env python3
"""Times the contraction engine on the EXACT launch shapes of one unit step (C2: B=64, N=20, L=3, T=15, D=768), each as
its launch site batches it (pre-split bf16 planes, several problems per launch), alone on the chip.  GPU only.
usage: python tools/gemm_shapes.py [prec=2] [reps=50]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L

lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B, N, T, D = 64, 20, 15, 768
BN, BT, HF = B * N, B * T, 1536


def planes(x):
    n = x.numel()
    pl = torch.empty(2 * n, device="cuda", dtype=torch.bfloat16)
    assert lib.dvgr_dbg_split(x.data_ptr(), pl.data_ptr(), n, n, st()) == 0
    return pl, n


def run(name, mode, M, Nn, K, count, tag, kseg=0, act=0, acc=0, atomic=0, rowsum=0):
    act |= int(os.environ.get("DVGR_GEMM_DBG", "0")) << 4      # timing aids of the planes kernel: 4 = no compute, 8 = no DMA
    shp = {0: ((M, K), (Nn, K)), 1: ((M, K), (K, Nn)), 2: ((K, M), (K, Nn))}[mode]
    A, Bm = torch.randn(*shp[0], device="cuda"), torch.randn(*shp[1], device="cuda")
    pa, na = planes(A)
    pb, nb = planes(Bm)
    Cm = torch.zeros(count * M, Nn, device="cuda")
    rs = torch.zeros(count * M, device="cuda")

    def call():
        rc = lib.dvgr_dbg_gemm_planes_batch(mode, M, Nn, K, kseg, A.data_ptr(), pa.data_ptr(), A.stride(0), na, Bm.data_ptr(),
                                            pb.data_ptr(), Bm.stride(0), nb, Cm.data_ptr(), Nn, act, acc, atomic, prec, count, tag,
                                            rs.data_ptr() if rowsum else None, 0, st())
        assert rc == 0, rc
    for _ in range(5):
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
    tiles = count * ((M + 127) // 128) * ((Nn + 127) // 128)
    print(f"{name:44s} mode {mode} {count} x ({M:5d} x {Nn:5d} x {K:5d})  tiles {tiles:4d}  {us:7.1f} us  {gf / us * 1e3:7.1f} TFLOP/s (algorithmic)")


if __name__ == "__main__":
    print(f"prec {prec} (0 fp32, 1 bf16, 2 bf16x3), {reps} launches back to back each")
    run("S2  fwd GAT  Wh = xd.Wcat^T (train: 4 x 768 cols)", 0, BN, 768, D, 4, 1)
    run("S2  fwd GAT  eval form (2 x 1536 cols)", 0, BN, HF, D, 2, 1)
    run("S4  fwd SF   tanh(z.P1^T)", 0, BN, D, D, 4, 2, act=1)
    run("S0  fwd QA   feat_enhance, 3 layers", 0, BT, D, D, 3, 0)
    run("S6  fwd MFB  x0, x1 (ELU)", 0, BN, 512, D, 2, 0, act=2)
    run("S6  fwd MFB  out (K = 256)", 0, BN, D, 256, 1, 0, act=2)
    run("bwd SF  dz += dpre.P1 (M = 2 B N)", 1, 2 * BN, D, D, 2, 6, acc=1)
    run("bwd GAT dX += dWh.Wcat (train: 4 x K 768)", 1, BN, D, 768, 4, 4, acc=1)
    run("bwd GAT dX eval form (2 x K 1536)", 1, BN, D, HF, 2, 4, acc=1)
    run("bwd GAT dWcat = dWh^T.xd (train: 4 x 768 x 768)", 2, 768, D, BN, 4, 3, atomic=1)
    run("bwd GAT dWcat eval form (2 x 1536 x 768)", 2, HF, D, BN, 2, 3, atomic=1)
    run("bwd SF  dP1 = dpre^T.z (2 segments)", 2, D, D, 2 * BN, 2, 5, kseg=BN, atomic=1)
    run("bwd QA  dW1 = dy^T.dq", 2, D, D, BT, 1, 0, atomic=1)
    run("bwd QA  d(dq) += dy.W1", 1, BT, D, D, 1, 0, acc=1, atomic=1)
    run("bwd MFB dA, dM (K = 512)", 1, BN, D, 512, 2, 0)
    run("bwd MFB dW0, dW1", 2, 512, D, BN, 2, 0, atomic=1)
    run("dW all of one layer in ONE launch (6 problems)", 2, 768, D, BN, 6, 3, atomic=1)
    run("  ... with the bias gradients (row sums) fused", 2, 768, D, BN, 6, 3, atomic=1, rowsum=1)
