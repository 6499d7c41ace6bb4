#!/usr/bin/env python3
"""Golden vectors of the TRAINING STEP around the path, recorded by IMPORTING the reference
(build container only; nothing is copied): two iterations of train.py:139-159 on the reference `DualVGR` --

    optimizer.zero_grad(); outputs = model(*batch); loss = CE + alpha * common / L + beta * dependence / L (utils.py:10-31);
    loss.backward(); nn.utils.clip_grad_norm_(model.parameters(), max_norm); optimizer.step()   # optim.Adam(params, lr)

-- in training mode (BatchNorm uses batch statistics) with every dropout probability patched to 0, so that the step is
deterministic.  Recorded: the gradients of the first step (first 2048 elements per parameter); per step: loss, its three terms, the total gradient norm clip_grad_norm_ returned; after the last
step: every parameter and BatchNorm buffer (first 2048 elements + sum and sum of magnitudes).  Two cases: max_norm = 12 (train.py:158) and a
large max_norm that leaves the gradient unclipped (at 12 the clip bites in both steps: norms 22.8, 21.0).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_step.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn.functional as F

import make_golden_model as MG          # shims (device strings -> CPU), the imported reference, seeded_state, model_inputs
import importlib.util as _ilu

_spec = _ilu.spec_from_file_location("ref_top_utils", os.path.join(MG.REF, "utils.py"))
ref_utils = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(ref_utils)

from dualvgr_videoqa_amd import synth  # noqa: E402

F.dropout = lambda x, p=0.5, training=True, inplace=False: x          # every dropout site: p = 0


A_SCALE = 20.0


def step_state(model, seed):
    """make_golden_model.seeded_state with the GAT attention vectors `a.weight` scaled by A_SCALE.  With xavier-sized `a`
    the logits LeakyReLU(u_i + v_j + c) of most samples sit on ONE side of the kink for every j, where softmax_j does not
    depend on i at all: every node then aggregates the same average, the rows of a GAT output are identical up to
    rounding, and common_loss -- which L2-normalises the rows AFTER centring them over the nodes (utils.py:11-14) --
    normalises pure rounding noise: its value then differs between any two implementations (and between two BLAS builds of
    the reference) in the first digit.  Larger logits straddle the kink, rows differ (smallest centred row norm 2.5e-3 of
    0.5 here), and the term is a meaningful golden value.  (seed and N = 8 were picked by scanning for that.)"""
    sd = MG.seeded_state(model, seed)
    for k in sd:
        if k.endswith(".a.weight"):
            sd[k] = sd[k] * A_SCALE
    return sd


def run_case(prefix, out, *, V, D, Wd, N, Fr, L, B, T, vocab_q, vocab_a, lr, alpha, beta, max_norm, steps, seed=224, full=True):
    vocab = {"question_token_to_idx": {str(i): i for i in range(vocab_q)}, "answer_token_to_idx": {str(i): i for i in range(vocab_a)}}
    model = MG.ref_models.DualVGR(vision_dim=V, module_dim=D, word_dim=Wd, vocab=vocab, num_of_nodes=N, graph_module="GAT",
                                  graph_layers=1, unit_layers=L)
    model.load_state_dict(step_state(model, seed))
    model.train()
    app, mot, q, qlen = MG.model_inputs(B, N, Fr, V, T, vocab_q, seed)
    answers = torch.from_numpy((synth.uniform("s_ans", (B,), 0.0, float(vocab_a), seed)).astype(np.int64))
    batch = [torch.from_numpy(x) for x in (app, mot, q, qlen)]
    opt = torch.optim.Adam(model.parameters(), lr)                                     # train.py:85
    crit = torch.nn.CrossEntropyLoss()                                                 # train.py:93
    rec = {k: [] for k in ("loss", "ce", "common", "dep", "norm")}
    for _ in range(steps):
        opt.zero_grad()                                                                # train.py:139
        logits, aq_embed, mq_embed, com_app, com_motion, aq_fusion, mq_fusion = model(*batch)
        ce = crit(logits, answers)                                                     # :147
        loss_dep, loss_com, temp = 0, 0, len(aq_fusion)                                # :148-154
        for i in range(temp):
            loss_dep = loss_dep + (ref_utils.loss_dependence(aq_fusion[i], com_app[i], N) +
                                   ref_utils.loss_dependence(mq_fusion[i], com_motion[i], N))
            loss_com = loss_com + ref_utils.common_loss(com_app[i], com_motion[i])
        loss = ce + alpha * loss_com / temp + beta * loss_dep / temp
        loss.backward()                                                                # :155
        if len(rec["loss"]) == 0 and full:                                             # first step: the raw gradients
            for k, p_ in model.named_parameters():
                g = p_.grad if p_.grad is not None else torch.zeros_like(p_)
                out[prefix + "g0/" + k] = g.numpy().reshape(-1)[:2048].copy()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)   # :158
        opt.step()                                                                     # :159
        for k, v in (("loss", loss), ("ce", ce), ("common", loss_com / temp), ("dep", loss_dep / temp), ("norm", norm)):
            rec[k].append(float(v))
    sd = model.state_dict()
    out[prefix + "meta"] = np.array([V, D, Wd, N, Fr, L, B, T, vocab_q, vocab_a, seed, steps], dtype=np.int64)
    out[prefix + "meta_f"] = np.array([lr, alpha, beta, max_norm, A_SCALE], dtype=np.float64)
    for k, v in rec.items():
        out[prefix + k] = np.array(v, dtype=np.float64)
    out[prefix + "keys"] = np.array(sorted(sd.keys()))
    out[prefix + "shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    for k in sorted(sd.keys()):                       # whole tensor up to 2048 elements, else its first 2048 + two checksums
        v = sd[k].numpy().reshape(-1)
        if not full and v.size > 1024:                # the second case keeps the small tensors only
            continue
        out[prefix + "after/" + k] = v[:2048].copy()
        out[prefix + "sum/" + k] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
    print(prefix, {k: [f"{x:.6g}" for x in v] for k, v in rec.items()})


if __name__ == "__main__":
    torch.set_num_threads(4)
    out = {}
    dims = dict(V=32, D=64, Wd=32, N=8, Fr=4, L=2, B=6, T=7, vocab_q=30, vocab_a=12)
    run_case("a/", out, **dims, lr=1e-3, alpha=1.0, beta=10.0, max_norm=12.0, steps=2)
    run_case("b/", out, **dims, lr=1e-3, alpha=1.0, beta=10.0, max_norm=100.0, steps=2, full=False)
    np.savez_compressed(os.path.join(HERE, "g_step_small.npz"), **out)
