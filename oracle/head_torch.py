"""ORACLE (test infrastructure, not product code): CPU restatement of the step after the unit --
`ContextSelfAttn` (model/AnswerDecoder.py:155-182) and `SimpleOutputUnitOpenEnded` (:184-202) as
`DualVGR.forward` chains them (model/models.py:80-81) -- over the reference's own state_dict keys.

Pinned by tests/golden/g_head_*.npz (outputs and gradients of the imported reference classes,
tests/golden/make_golden_head.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
may import this module.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
DROP_P = 0.15          # model/AnswerDecoder.py:163, :190, :194
BN_EPS = 1e-5          # nn.BatchNorm1d default (model/AnswerDecoder.py:193)
BN_MOMENTUM = 0.1

PREFIX_A = "feature_aggregation."      # names as in DualVGR (model/models.py:38, :49)
PREFIX_O = "output_unit."


def param_shapes(module_dim: int, num_answers: int) -> Dict[str, tuple]:
    D, C = module_dim, num_answers
    return {
        PREFIX_A + "v_proj.weight": (D, D),
        PREFIX_A + "attn.weight": (1, D),
        PREFIX_A + "attn.bias": (1,),
        PREFIX_O + "question_proj.weight": (D, D),
        PREFIX_O + "question_proj.bias": (D,),
        PREFIX_O + "classifier.1.weight": (D, 2 * D),
        PREFIX_O + "classifier.1.bias": (D,),
        PREFIX_O + "classifier.3.weight": (D,),
        PREFIX_O + "classifier.3.bias": (D,),
        PREFIX_O + "classifier.5.weight": (C, D),
        PREFIX_O + "classifier.5.bias": (C,),
    }


def _drop(x: Tensor, mask: Optional[Tensor], train: bool) -> Tensor:
    """nn.Dropout(0.15): identity in eval; in training the injected keep-mask (1 = keep)."""
    if not train:
        return x
    assert mask is not None, "training mode needs the keep-masks (they are injected, never drawn here)"
    return x * mask / (1.0 - DROP_P)


def head_forward(sd: Dict[str, Tensor], visual_feature: Tensor, question_embedding: Tensor, train: bool = False,
                 masks: Optional[Dict[str, Tensor]] = None, running: Optional[Dict[str, Tensor]] = None):
    """Returns (logits (B,C), aux) with aux = {'attn' (B,N), 'v_distill' (B,D), 'bn_mean', 'bn_var' (biased),
    'new_running_mean', 'new_running_var'}.

    masks: 'agg' (B,N,D), 'cat' (B,2D), 'bn' (B,D) keep-masks of the three nn.Dropout sites, in call order.
    running: {'mean','var'} BatchNorm1d running statistics (eval mode normalises with them)."""
    masks = masks or {}
    A, Oq = PREFIX_A, PREFIX_O
    # ---- ContextSelfAttn.forward, model/AnswerDecoder.py:173-182 ----
    x = _drop(visual_feature, masks.get("agg"), train)                       # :173
    v_proj = x @ sd[A + "v_proj.weight"].t()                                   # :174 (bias=False, :159)
    e = F.elu(v_proj)                                                          # :175
    attn = e @ sd[A + "attn.weight"].t() + sd[A + "attn.bias"]                 # :177  (B,N,1)
    attn = torch.softmax(attn, dim=1)                                          # :178  over the clips
    v_distill = (attn * x).sum(1)                                              # :180  weights the DROPPED features
    # ---- SimpleOutputUnitOpenEnded.forward, model/AnswerDecoder.py:197-202 ----
    q = question_embedding @ sd[Oq + "question_proj.weight"].t() + sd[Oq + "question_proj.bias"]   # :198
    out = torch.cat([v_distill, q], 1)                                         # :199
    out = _drop(out, masks.get("cat"), train)                                  # classifier.0
    h = out @ sd[Oq + "classifier.1.weight"].t() + sd[Oq + "classifier.1.bias"]
    h = F.elu(h)                                                               # classifier.2
    aux = {"attn": attn.squeeze(-1), "v_distill": v_distill}
    if train:                                                                  # classifier.3: batch statistics
        mean = h.mean(0)
        var = h.var(0, unbiased=False)
        n = h.shape[0]
        aux["bn_mean"], aux["bn_var"] = mean, var
        if running is not None:
            aux["new_running_mean"] = (1 - BN_MOMENTUM) * running["mean"] + BN_MOMENTUM * mean.detach()
            aux["new_running_var"] = (1 - BN_MOMENTUM) * running["var"] + BN_MOMENTUM * var.detach() * n / max(n - 1, 1)
    else:
        assert running is not None, "eval mode normalises with the running statistics"
        mean, var = running["mean"], running["var"]
    y = (h - mean) / torch.sqrt(var + BN_EPS) * sd[Oq + "classifier.3.weight"] + sd[Oq + "classifier.3.bias"]
    y = _drop(y, masks.get("bn"), train)                                       # classifier.4
    logits = y @ sd[Oq + "classifier.5.weight"].t() + sd[Oq + "classifier.5.bias"]
    return logits, aux
