"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PyTorch-CPU fp32 restatement of the
DualVGR reasoning unit, written functionally over a reference-compatible ``state_dict``.

Every function cites the reference lines (relative to /root/reference) whose arithmetic it
restates.  Two flavours of the punish-GAT head exist:

* ``naive=True``  -- shape-faithful to the reference: materialises the (B, N*N, F) "chunks" and
  "alternating" copies and their (B, N, N, 2F) concatenation before the ``a`` projection
  (model/GraphNN.py:115-155).  This is what ``bench.py`` times as the CPU baseline.
* ``naive=False`` -- the factorised form e_ij = LeakyReLU(a_l.Wh_i + a_r.Wh_j + c) that the HIP
  kernels implement.

Dropout is reproduced by *mask injection*: ``masks`` maps a site name to a 0/1 keep tensor and
the kept values are scaled by 1/(1-p), which is what ``F.dropout`` does (model/GraphNN.py:107,
175,177; p = 0.15 from model/models.py:95-100).  ``masks=None`` is eval mode.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
DROP_P = 0.15          # model/models.py:95-100 (dropout=0.15 for every punishGAT)
LRELU_SLOPE = 0.01     # model/models.py:95-100 (alpha=0.01)
N_HEADS = 4            # model/models.py:95-100 (n_heads=4)
GAT_FAMILIES = ("acGCN", "appearance_GCN", "mcGCN", "motion_GCN")


def _linear(x: Tensor, sd: Dict[str, Tensor], key: str) -> Tensor:
    b = sd.get(key + ".bias")
    return F.linear(x, sd[key + ".weight"], b)


def _drop(x: Tensor, masks: Optional[Dict[str, Tensor]], site: str) -> Tensor:
    if masks is None:
        return x
    keep = masks[site].to(x.dtype)
    return x * keep / (1.0 - DROP_P)


def query_attn(sd, prefix: str, words: Tensor, dq: Tensor, qlen: Tensor) -> Tuple[Tensor, Tensor]:
    """model/utils.py:66-84 (QueryAttn.forward)."""
    y = _linear(dq, sd, prefix + ".feat_enhance")                       # :68
    d = F.normalize(y, p=2, dim=-1)                                     # :68 (eps 1e-12)
    r = _linear(d, sd, prefix + ".fc").squeeze(2)                       # :69
    a = torch.softmax(r, dim=1)                                         # :69 softmax over ALL T
    T = dq.shape[1]
    valid = (torch.arange(T, device=dq.device)[None, :] < qlen.to(dq.device)[:, None]).to(a.dtype)  # :72-75
    a = a * valid                                                       # :76
    a = a / (a.sum(1, keepdim=True) + 1e-5)                             # :77
    q = torch.bmm(a.unsqueeze(1), words).squeeze(1)                     # :79-82
    return q, a


def query_punish(sd, prefix: str, q: Tensor, x: Tensor) -> Tensor:
    """model/utils.py:92-105 (QueryPunish.forward); returns the (B, N, 1) gate before expand."""
    p = _linear(q, sd, prefix + ".query_weight")                        # :101
    return torch.sigmoid(torch.bmm(x, p.unsqueeze(2)))                  # :102-103


def gat_head(sd, prefix: str, x: Tensor, s: Tensor, masks, site: str, naive: bool) -> Tensor:
    """model/GraphNN.py:95-113 (PunishGraphAttentionLayer.forward), adjacency mask is a no-op
    because the normalised all-ones adjacency is strictly positive (model/models.py:114-116)."""
    Wh = _linear(x, sd, prefix + ".W")                                  # :96
    B, N, Fh = Wh.shape
    a_w = sd[prefix + ".a.weight"]
    a_b = sd[prefix + ".a.bias"]
    if naive:
        chunks = Wh.repeat_interleave(N, dim=1)                         # :130 e1,e1,..,e2,e2,..
        altern = Wh.repeat(1, N, 1)                                     # :131 e1,e2,..,e1,e2,..
        pairs = torch.cat([chunks, altern], dim=2).view(B, N, N, 2 * Fh)  # :152-155
        pre = F.linear(pairs, a_w, a_b).squeeze(-1)                     # :98
    else:
        u = Wh @ a_w[0, :Fh]
        v = Wh @ a_w[0, Fh:]
        pre = u[:, :, None] + v[:, None, :] + a_b
    e = F.leaky_relu(pre, LRELU_SLOPE)                                  # :98
    Z = Wh * s                                                          # :102-104 (after logits)
    alpha = torch.softmax(e, dim=-1)                                    # :106
    alpha = _drop(alpha, masks, site + ".alpha")                        # :107
    return F.elu(torch.bmm(alpha, Z))                                   # :108-111


def punish_gat(sd, prefix: str, x: Tensor, s: Tensor, masks, site: str, naive: bool) -> Tensor:
    """model/GraphNN.py:174-178 (punishGAT.forward)."""
    x = _drop(x, masks, site + ".x")                                    # :175
    heads = [gat_head(sd, f"{prefix}.attention_{h}", x, s, masks, f"{site}.h{h}", naive)
             for h in range(N_HEADS)]
    out = torch.cat(heads, dim=2)                                       # :176
    return _drop(out, masks, site + ".out")                             # :177


def sfgcn_attention(sd, prefix: str, z: Tensor) -> Tuple[Tensor, Tensor]:
    """model/Attention.py:20-23 (AttentionSFGCN.forward); z is (B, 2, N, D)."""
    t = torch.tanh(_linear(z, sd, prefix + ".project.0"))
    w = F.linear(t, sd[prefix + ".project.2.weight"])
    beta = torch.softmax(w, dim=1)
    return (beta * z).sum(1), beta


def mfb(sd, prefix: str, a: Tensor, m: Tensor) -> Tensor:
    """model/fusions/fusions.py:419-453 with the call-site configuration model/models.py:109
    (mm_dim 256, factor 2, ELU in/out, no normalisation, no dropout)."""
    x0 = F.elu(_linear(a, sd, prefix + ".linear0"))
    x1 = F.elu(_linear(m, sd, prefix + ".linear1"))
    z = x0 * x1
    z = z.view(*z.shape[:-1], z.shape[-1] // 2, 2).sum(-1)            # :439-440 adjacent pairs
    return F.elu(_linear(z, sd, prefix + ".linear_out"))


def unit_forward(sd: Dict[str, Tensor], A: Tensor, M: Tensor, dq: Tensor, words: Tensor,
                 qlen: Tensor, unit_layers: int, graph_layers: int = 1,
                 masks: Optional[Dict[str, Tensor]] = None, naive: bool = True):
    """model/models.py:121-173 (DualVGRUnit_multiple.forward).  Returns the same 7-tuple; the
    four lists stay on the input device (the reference moves them to CPU, :153-160)."""
    aq_list: List[Tensor] = []
    mq_list: List[Tensor] = []
    ca_list: List[Tensor] = []
    cm_list: List[Tensor] = []
    aq_embed = mq_embed = None
    for i in range(unit_layers):
        aq, mq = A, M
        q, _ = query_attn(sd, f"queryAttn.{i}", words, dq, qlen)        # :145
        sA = query_punish(sd, f"queryPunish_appear.{i}", q, aq)         # :146
        sM = query_punish(sd, f"queryPunish_motion.{i}", q, mq)         # :147
        for j in range(graph_layers):                                   # :150-154
            k = i + j                                                   # reference indexes [i+j]
            com_a = punish_gat(sd, f"acGCN.{k}", aq, sA, masks, f"L{i}.G{j}.acGCN", naive)
            aq = punish_gat(sd, f"appearance_GCN.{k}", aq, sA, masks, f"L{i}.G{j}.appearance_GCN", naive)
            aq_list.append(aq)
            ca_list.append(com_a)
        for j in range(graph_layers):                                   # :156-160
            k = i + j
            com_m = punish_gat(sd, f"mcGCN.{k}", mq, sM, masks, f"L{i}.G{j}.mcGCN", naive)
            mq = punish_gat(sd, f"motion_GCN.{k}", mq, sM, masks, f"L{i}.G{j}.motion_GCN", naive)
            mq_list.append(mq)
            cm_list.append(com_m)
        aq_embed, _ = sfgcn_attention(sd, f"attention_appearance.{i}", torch.stack([com_a, aq], 1))  # :163-165
        mq_embed, _ = sfgcn_attention(sd, f"attention_motion.{i}", torch.stack([com_m, mq], 1))      # :164,166
        A = A + aq_embed                                                # :168
        M = M + mq_embed                                                # :169
    vis = mfb(sd, "visualfusion", A, M)                                 # :171
    return vis, aq_embed, mq_embed, ca_list, cm_list, aq_list, mq_list


# --------------------------------------------------------------------------------------------
# Auxiliary losses that consume the unit's list outputs (top-level utils.py of the reference).
# --------------------------------------------------------------------------------------------
def common_loss(e1: Tensor, e2: Tensor) -> Tensor:
    """utils.py:10-18."""
    e1 = F.normalize(e1 - e1.mean(dim=1, keepdim=True), p=2, dim=2)
    e2 = F.normalize(e2 - e2.mean(dim=1, keepdim=True), p=2, dim=2)
    c1 = e1 @ e1.transpose(1, 2)
    c2 = e2 @ e2.transpose(1, 2)
    return ((c1 - c2) ** 2).mean()


def loss_dependence(e1: Tensor, e2: Tensor, dim: int) -> Tensor:
    """utils.py:20-31 (HSIC, un-normalised SUM over the batch)."""
    R = torch.eye(dim, dtype=e1.dtype, device=e1.device) - 1.0 / dim
    K1 = e1 @ e1.transpose(1, 2)
    K2 = e2 @ e2.transpose(1, 2)
    prod = (R @ K1) @ (R @ K2)
    return torch.diagonal(prod, dim1=1, dim2=2).sum()


def aux_losses(outs, num_nodes: int) -> Tuple[Tensor, Tensor]:
    """train.py:148-154: (sum_l common_l)/L' and (sum_l dep_l)/L' over the four lists."""
    _, _, _, ca, cm, aq, mq = outs
    n = len(aq)
    com = sum(common_loss(ca[l], cm[l]) for l in range(n)) / n
    dep = sum(loss_dependence(aq[l], ca[l], num_nodes) + loss_dependence(mq[l], cm[l], num_nodes)
              for l in range(n)) / n
    return com, dep


def bench_scalar_loss(outs, num_nodes: int, alpha: float = 1.0, beta: float = 1e-8) -> Tensor:
    """Scope-U loss of SURVEY.md section 8(d): sum(visual_feature) + alpha*common + beta*HSIC."""
    com, dep = aux_losses(outs, num_nodes)
    return outs[0].sum() + alpha * com + beta * dep


# --------------------------------------------------------------------------------------------
# Parameter inventory (state_dict keys/shapes of DualVGRUnit_multiple)
# --------------------------------------------------------------------------------------------
def param_shapes(word_dim: int, module_dim: int, unit_layers: int, graph_layers: int = 1):
    D, Wd, Fh = module_dim, word_dim, module_dim // 4
    shapes = {}
    for i in range(unit_layers):
        shapes[f"queryAttn.{i}.feat_enhance.weight"] = (D, D)
        shapes[f"queryAttn.{i}.feat_enhance.bias"] = (D,)
        shapes[f"queryAttn.{i}.fc.weight"] = (1, D)
        shapes[f"queryAttn.{i}.fc.bias"] = (1,)
        for fam in ("queryPunish_appear", "queryPunish_motion"):
            shapes[f"{fam}.{i}.query_weight.weight"] = (D, Wd)
            shapes[f"{fam}.{i}.query_weight.bias"] = (D,)
        for fam in ("attention_appearance", "attention_motion"):
            shapes[f"{fam}.{i}.project.0.weight"] = (D, D)
            shapes[f"{fam}.{i}.project.0.bias"] = (D,)
            shapes[f"{fam}.{i}.project.2.weight"] = (1, D)
    for k in range(unit_layers * graph_layers):
        for fam in GAT_FAMILIES:
            for h in range(N_HEADS):
                shapes[f"{fam}.{k}.attention_{h}.W.weight"] = (Fh, D)
                shapes[f"{fam}.{k}.attention_{h}.W.bias"] = (Fh,)
                shapes[f"{fam}.{k}.attention_{h}.a.weight"] = (1, 2 * Fh)
                shapes[f"{fam}.{k}.attention_{h}.a.bias"] = (1,)
    shapes["visualfusion.linear0.weight"] = (512, D)
    shapes["visualfusion.linear0.bias"] = (512,)
    shapes["visualfusion.linear1.weight"] = (512, D)
    shapes["visualfusion.linear1.bias"] = (512,)
    shapes["visualfusion.linear_out.weight"] = (D, 256)
    shapes["visualfusion.linear_out.bias"] = (D,)
    return shapes


def dropout_site_shapes(B: int, N: int, D: int, unit_layers: int, graph_layers: int = 1):
    """Names/shapes of every Bernoulli site of one training forward (24 per layer at G=1)."""
    sites = {}
    for i in range(unit_layers):
        for j in range(graph_layers):
            for fam in GAT_FAMILIES:
                base = f"L{i}.G{j}.{fam}"
                sites[base + ".x"] = (B, N, D)
                sites[base + ".out"] = (B, N, D)
                for h in range(N_HEADS):
                    sites[f"{base}.h{h}.alpha"] = (B, N, N)
    return sites
