#!/usr/bin/env python3
"""Golden vectors of the answer head (row f2): outputs and gradients of the reference's `ContextSelfAttn` and
`SimpleOutputUnitOpenEnded` classes, IMPORTED from /root/reference (build container only; nothing is copied).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_head.py

Training-mode cases replace `F.dropout` by a recording version so that the Bernoulli keep-masks the reference drew can
be injected into the oracle and replayed by the HIP path (same approach as make_golden.py)."""
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
import model.AnswerDecoder as ref_dec  # noqa: E402  (the reference, imported in place)

from dualvgr_videoqa_amd import synth  # noqa: E402
from oracle import head_torch as H  # noqa: E402

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


def run_case(name, *, D, C, N, B, train, seed=777):
    shapes = H.param_shapes(D, C)
    sd_np = synth.xavier_state_dict(shapes, seed=seed, bias_range=0.1)
    # BatchNorm affine parameters and running statistics away from their defaults
    sd_np[H.PREFIX_O + "classifier.3.weight"] = (1.0 + 0.2 * synth.normal("bn_w", (D,), 1.0, seed)).astype(np.float32)
    sd_np[H.PREFIX_O + "classifier.3.bias"] = (0.1 * synth.normal("bn_b", (D,), 1.0, seed)).astype(np.float32)
    run_mean = (0.05 * synth.normal("bn_rm", (D,), 1.0, seed)).astype(np.float32)
    run_var = (1.0 + 0.3 * synth.uniform("bn_rv", (D,), 0.0, 1.0, seed)).astype(np.float32)
    vis = synth.normal("head_vis", (B, N, D), 0.7, seed)
    qe = synth.normal("head_qe", (B, D), 0.5, seed)
    Rl = synth.normal("head_R", (B, C), 1.0, seed)

    agg = ref_dec.ContextSelfAttn(D)
    outu = ref_dec.SimpleOutputUnitOpenEnded(D, C)
    agg.load_state_dict({k[len(H.PREFIX_A):]: torch.from_numpy(v.copy()) for k, v in sd_np.items() if k.startswith(H.PREFIX_A)})
    osd = {k[len(H.PREFIX_O):]: torch.from_numpy(v.copy()) for k, v in sd_np.items() if k.startswith(H.PREFIX_O)}
    osd["classifier.3.running_mean"] = torch.from_numpy(run_mean.copy())
    osd["classifier.3.running_var"] = torch.from_numpy(run_var.copy())
    osd["classifier.3.num_batches_tracked"] = torch.tensor(0)
    outu.load_state_dict(osd)
    agg.train(train)
    outu.train(train)
    rec = DropoutRecorder()
    if train:
        torch.manual_seed(seed)
        F.dropout = rec
        torch.nn.functional.dropout = rec
    v = torch.from_numpy(vis.copy()).requires_grad_(True)
    q = torch.from_numpy(qe.copy()).requires_grad_(True)
    vd = agg(v)                                  # model/models.py:80
    logits = outu(q, vd)                         # model/models.py:81
    loss = (logits * torch.from_numpy(Rl)).sum()
    loss.backward()
    F.dropout = _REAL_DROPOUT
    torch.nn.functional.dropout = _REAL_DROPOUT
    out = {"meta": np.array([D, C, N, B, int(train), seed], dtype=np.int64),
           "run_mean": run_mean, "run_var": run_var,
           "logits": logits.detach().numpy(), "v_distill": vd.detach().numpy(),
           "g_vis": v.grad.numpy(), "g_qe": q.grad.numpy(),
           "new_run_mean": outu.classifier[3].running_mean.numpy().copy(),
           "new_run_var": outu.classifier[3].running_var.numpy().copy()}
    grads = {H.PREFIX_A + k: p.grad.numpy() for k, p in agg.named_parameters()}
    grads.update({H.PREFIX_O + k: p.grad.numpy() for k, p in outu.named_parameters()})
    names = sorted(grads)
    out["gw_names"] = np.array(names)
    out["gw_sum"] = np.array([grads[k].astype(np.float64).sum() for k in names])
    out["gw_abs"] = np.array([np.abs(grads[k].astype(np.float64)).sum() for k in names])
    out["gw_head"] = np.stack([np.resize(grads[k].ravel()[:16], 16) for k in names])
    if train:
        assert len(rec.masks) == 3, len(rec.masks)
        for nm, mk in zip(("agg", "cat", "bn"), rec.masks):
            out["mask/" + nm] = np.packbits(mk.numpy().astype(np.uint8).ravel())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "logits", float(np.abs(out["logits"]).max()), "masks", len(rec.masks))


if __name__ == "__main__":
    run_case("g_head_small_eval", D=64, C=24, N=5, B=6, train=False)
    run_case("g_head_small_train", D=64, C=24, N=5, B=6, train=True)
    run_case("g_head_real_eval", D=768, C=1000, N=20, B=16, train=False)
    run_case("g_head_real_train", D=768, C=1000, N=20, B=16, train=True)
    run_case("g_head_n1_b3_train", D=64, C=8, N=1, B=3, train=True)
