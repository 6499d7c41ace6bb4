#!/usr/bin/env python3
"""Golden vectors of the question encoder (row f3): outputs and gradients of the reference's `InputUnitLinguisticDynamic`
class (packed BiLSTMs, model/Preprocessing.py:89-127), IMPORTED from /root/reference (build container only; nothing is
copied).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_question.py

Training-mode cases record the keep-masks the reference drew (as make_golden_clip.py)."""
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


def case_tensors(VQ, Wd, D, B, T, seed):
    sd = synth.xavier_state_dict(CT.question_param_shapes(VQ, Wd, D), seed=seed, bias_range=0.1)
    qlen = (1 + (synth.uniform("q_len", (B,), 0.0, 1.0, seed) * T).astype(np.int64)).clip(1, T)
    qlen[0] = T
    if B > 2:
        qlen[1] = 1
    q = (2 + (synth.uniform("q_tok", (B, T), 0.0, 1.0, seed) * (VQ - 2)).astype(np.int64)).clip(2, VQ - 1)
    q = q * (np.arange(T)[None, :] < qlen[:, None])
    R = {"qe": synth.normal("q_Rqe", (B, D), 1.0, seed), "words": synth.normal("q_Rw", (B, T, Wd), 1.0, seed),
         "seq": synth.normal("q_Rs", (B, T, D), 1.0, seed)}
    return sd, q.astype(np.int64), qlen.astype(np.int64), R


def run_case(name, *, VQ, Wd, D, B, T, train, seed=5151):
    sd_np, q, qlen, R = case_tensors(VQ, Wd, D, B, T, seed)
    enc = ref_pre.InputUnitLinguisticDynamic(vocab_size=VQ, wordvec_dim=Wd, rnn_dim=D)
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    enc.train(train)
    rec = DropoutRecorder()
    if train:
        torch.manual_seed(seed)
        F.dropout = rec
        torch.nn.functional.dropout = rec
    qe, words, seq = enc(torch.from_numpy(q), torch.from_numpy(qlen))
    loss = (qe * torch.from_numpy(R["qe"])).sum() + (words * torch.from_numpy(R["words"])).sum() + \
        (seq * torch.from_numpy(R["seq"])).sum()
    loss.backward()
    F.dropout = _REAL_DROPOUT
    torch.nn.functional.dropout = _REAL_DROPOUT
    z = {"meta": np.array([VQ, Wd, D, B, T, int(train), seed], dtype=np.int64), "qe": qe.detach().numpy(),
         "words": words.detach().numpy(), "seq": seq.detach().numpy()}
    for k, p in enc.named_parameters():
        z["g_" + k] = p.grad.numpy()
    if train:
        assert len(rec.masks) == 2
        z["mask_emb"] = np.packbits(rec.masks[0].numpy().astype(np.uint8).reshape(-1))
        z["mask_final"] = np.packbits(rec.masks[1].numpy().astype(np.uint8).reshape(-1))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **z)
    print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    run_case("g_question_small_eval", VQ=40, Wd=20, D=64, B=5, T=7, train=False)
    run_case("g_question_small_train", VQ=40, Wd=20, D=64, B=4, T=6, train=True)
    run_case("g_question_wd300_eval", VQ=60, Wd=300, D=64, B=3, T=9, train=False)
