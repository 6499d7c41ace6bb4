#!/usr/bin/env python3
"""Times the appearance encoder (forward + backward, training mode) at a bench shape: the HIP ClipEncoder against the
stock nn.LSTM wrapper (MIOpen) of tools/stock_encoders.py.  usage: python tools/clip_time.py [B N T V D dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dualvgr_videoqa_amd as dv
from tools.stock_encoders import VisualAppearanceEncoder

a = sys.argv[1:]
B, N, T, V, D = [int(v) for v in (a[:5] if len(a) >= 5 else (64, 20, 16, 2048, 768))]
dtype = a[5] if len(a) > 5 else "bf16x3"
x = torch.rand(B, N, T, V, device="cuda")
R = torch.randn(B, N, D, device="cuda")


def run(m, iters=10):
    m.train()
    for _ in range(3):
        out = m(x); (out * R).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = m(x); (out * R).sum().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


hip = dv.ClipEncoder(V, D, dtype=dtype).cuda()
print(f"shape B={B} N={N} T={T} V={V} D={D} dtype={dtype}")
print(f"hip ClipEncoder      : {run(hip):8.3f} ms / step (fwd+bwd)")
if os.environ.get("CLIP_STOCK", "1") == "1":
    stock = VisualAppearanceEncoder(V, D).cuda()
    print(f"nn.LSTM (MIOpen)     : {run(stock):8.3f} ms / step (fwd+bwd)")
