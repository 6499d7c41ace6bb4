#!/usr/bin/env python3
"""Golden vectors of the appearance encoder (row f3): outputs and gradients of the reference's
`VisualAppearanceEncoder` class, IMPORTED from /root/reference (build container only; nothing is copied).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_clip.py

Training-mode cases replace `F.dropout` by a recording version so that the Bernoulli keep-masks the reference drew can
be injected into the oracle and replayed by the HIP path (same approach as make_golden.py).  Weights and inputs are
regenerated from the seed (dualvgr_videoqa_amd.synth); the big case stores strided samples of the weight gradients."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
sys.path.insert(0, REF)
import model.Preprocessing as ref_pre  # noqa: E402  (the reference, imported in place)

from dualvgr_videoqa_amd import synth  # noqa: E402
from oracle import clip_torch as CT  # noqa: E402

torch.set_num_threads(4)
_REAL_DROPOUT = F.dropout


class DropoutRecorder:
    def __init__(self):
        self.masks = []

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = torch.bernoulli(torch.full_like(x, 1.0 - p))
        self.masks.append(keep)
        return x * keep / (1.0 - p)


def case_tensors(V, D, B, N, T, seed):
    sd = synth.xavier_state_dict(CT.param_shapes(V, D), seed=seed, bias_range=0.1)
    x = np.abs(synth.normal("clip_x", (B, N, T, V), 0.8, seed)).astype(np.float32)     # ResNet features are >= 0
    R = synth.normal("clip_R", (B, N, D), 1.0, seed)
    return sd, x, R


def run_case(name, *, V, D, B, N, T, train, seed=4242, stride=1):
    sd_np, x, R = case_tensors(V, D, B, N, T, seed)
    enc = ref_pre.VisualAppearanceEncoder(appearance_dim=V, module_dim=D)
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    enc.train(train)
    rec = DropoutRecorder()
    if train:
        torch.manual_seed(seed)
        F.dropout = rec
        torch.nn.functional.dropout = rec
    out = enc(torch.from_numpy(x.copy()))
    (out * torch.from_numpy(R)).sum().backward()
    F.dropout = _REAL_DROPOUT
    torch.nn.functional.dropout = _REAL_DROPOUT
    z = {"meta": np.array([V, D, B, N, T, int(train), seed, stride], dtype=np.int64), "out": out.detach().numpy()}
    for k, p in enc.named_parameters():
        g = p.grad.numpy()
        z["g_" + k] = g[::stride, ::stride] if g.ndim == 2 else g
    if train:
        assert len(rec.masks) == 2
        z["mask_in"] = np.packbits(rec.masks[0].numpy().astype(np.uint8).reshape(-1))
        z["mask_out"] = np.packbits(rec.masks[1].numpy().astype(np.uint8).reshape(-1))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **z)
    print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    run_case("g_clip_small_eval", V=64, D=64, B=2, N=3, T=4, train=False)
    run_case("g_clip_small_train", V=64, D=64, B=2, N=3, T=5, train=True)
    run_case("g_clip_t1_eval", V=32, D=64, B=1, N=2, T=1, train=False)
    run_case("g_clip_real_train", V=2048, D=768, B=2, N=4, T=16, train=True, stride=16)
