#!/usr/bin/env python3
"""How many bf16 MFMA terms the 1e-4 forward gate needs: max |error| / max |output| of the S2 contraction (C2 shape,
N(0,1) activations, xavier-uniform weights) with the operands split into bf16 hi + lo planes, against float64."""
import numpy as np
import torch

rng = np.random.default_rng(0)
M, K, N = 1280, 768, 1536
x = rng.standard_normal((M, K)).astype(np.float32)
b = np.sqrt(6 / (K + N))
w = rng.uniform(-b, b, (N, K)).astype(np.float32)


def split(a):
    t = torch.from_numpy(a)
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi.numpy().astype(np.float64), lo.numpy().astype(np.float64)


xh, xl = split(x)
wh, wl = split(w)
ref = x.astype(np.float64) @ w.astype(np.float64).T
rel = lambda y: np.abs(y - ref).max() / np.abs(ref).max()
print(f"3 terms  hi.hi + hi.lo + lo.hi          {rel(xh @ wh.T + xh @ wl.T + xl @ wh.T):.2e}")
print(f"2 terms  hi.hi + hi.lo (x to bf16)      {rel(xh @ wh.T + xh @ wl.T):.2e}")
print(f"2 terms  hi.hi + lo.hi (w to bf16)      {rel(xh @ wh.T + xl @ wh.T):.2e}")
print(f"1 term   hi.hi                          {rel(xh @ wh.T):.2e}")
