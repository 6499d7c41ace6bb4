#!/usr/bin/env python3
"""Golden vectors of the whole `DualVGR` model (model/models.py:34-83), recorded by IMPORTING the reference (build
container only; nothing is copied): eval-mode logits and the six auxiliary outputs for seeded weights and inputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_model.py

Shims as in make_golden.py: `.to('cuda:1')` -> CPU, `.cuda()` -> identity (the reference hard-codes devices)."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

REF = "/root/reference"
_orig_to = torch.Tensor.to


def _to_cpu(self, *args, **kwargs):
    args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)
    if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
        kwargs["device"] = "cpu"
    return _orig_to(self, *args, **kwargs)


torch.Tensor.to = _to_cpu
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
import model.models as ref_models  # noqa: E402

from dualvgr_videoqa_amd import synth  # noqa: E402

torch.set_num_threads(4)


def model_inputs(B, N, Fr, V, T, vocab_q, seed):
    app = np.abs(synth.normal("m_app", (B, N, Fr, V), 0.5, seed))
    mot = np.abs(synth.normal("m_mot", (B, N, V), 0.5, seed))
    qlen = (3 + (synth.uniform("m_qlen", (B,), 0.0, 1.0, seed) * (T - 2)).astype(np.int64)).clip(1, T)
    qlen[0] = T
    q = (2 + (synth.uniform("m_q", (B, T), 0.0, 1.0, seed) * (vocab_q - 2)).astype(np.int64)).clip(2, vocab_q - 1)
    q = q * (np.arange(T)[None, :] < qlen[:, None])
    return app.astype(np.float32), mot.astype(np.float32), q.astype(np.int64), qlen.astype(np.int64)


def seeded_state(model, seed):
    sd = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if v.dtype != torch.float32:
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(1.0 + 0.3 * synth.uniform(k, shape, 0.0, 1.0, seed))
        elif len(shape) >= 2:
            bound = (6.0 / (shape[0] + shape[-1])) ** 0.5
            sd[k] = torch.from_numpy(synth.uniform(k, shape, -bound, bound, seed))
        else:
            sd[k] = torch.from_numpy(synth.uniform(k, shape, -0.1, 0.1, seed))
            if k.endswith("classifier.3.weight"):
                sd[k] = sd[k] + 1.0
    return sd


def run_case(name, *, V, D, Wd, N, Fr, L, B, T, vocab_q, vocab_a, seed=4242):
    vocab = {"question_token_to_idx": {str(i): i for i in range(vocab_q)}, "answer_token_to_idx": {str(i): i for i in range(vocab_a)}}
    model = ref_models.DualVGR(vision_dim=V, module_dim=D, word_dim=Wd, vocab=vocab, num_of_nodes=N, graph_module="GAT",
                               graph_layers=1, unit_layers=L)
    sd = seeded_state(model, seed)
    model.load_state_dict(sd)
    model.eval()
    app, mot, q, qlen = model_inputs(B, N, Fr, V, T, vocab_q, seed)
    with torch.no_grad():
        outs = model(torch.from_numpy(app), torch.from_numpy(mot), torch.from_numpy(q), torch.from_numpy(qlen))
    logits, aq_embed, mq_embed, com_app, com_mot, aq_fusion, mq_fusion = outs
    out = {"meta": np.array([V, D, Wd, N, Fr, L, B, T, vocab_q, vocab_a, seed], dtype=np.int64),
           "keys": np.array(sorted(sd.keys())),
           "shapes": np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())]),
           "logits": logits.numpy(), "aq_embed": aq_embed.numpy(), "mq_embed": mq_embed.numpy(),
           "com_app_last": com_app[-1].detach().numpy(), "mq_fusion_last": mq_fusion[-1].detach().numpy()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "logits max", float(np.abs(out["logits"]).max()), "params", len(sd))


if __name__ == "__main__":
    run_case("g_model_small", V=64, D=64, Wd=32, N=5, Fr=4, L=2, B=4, T=7, vocab_q=30, vocab_a=12)
