"""ORACLE (test infrastructure, not product code): CPU restatement of the appearance encoder in front of the unit --
`VisualAppearanceEncoder.forward` (model/Preprocessing.py:191-234) -- with the LSTM written out step by step
(torch.nn.LSTM's cell: gate order i, f, g, o, two bias vectors) over the reference's own state_dict keys, so the
keep-masks of the two dropout sites can be injected and autograd gives every gradient.

Pinned by tests/golden/g_clip_*.npz (outputs and gradients of the imported reference class, which runs nn.LSTM;
tests/golden/make_golden_clip.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

Tensor = torch.Tensor
DROP_IN = 0.15         # model/Preprocessing.py:199  embedding_dropout
DROP_OUT = 0.18        # model/Preprocessing.py:200  finalvisual_dropout
PREFIX = "encoder."    # names under DualVGR.visual_appearance_input_unit (model/models.py:43)


def param_shapes(appearance_dim: int, module_dim: int) -> Dict[str, tuple]:
    V, H = appearance_dim, module_dim // 2
    out = {}
    for sfx in ("", "_reverse"):
        out[PREFIX + "weight_ih_l0" + sfx] = (4 * H, V)
        out[PREFIX + "weight_hh_l0" + sfx] = (4 * H, H)
        out[PREFIX + "bias_ih_l0" + sfx] = (4 * H,)
        out[PREFIX + "bias_hh_l0" + sfx] = (4 * H,)
    return out


def _direction(embed: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor, reverse: bool) -> Tensor:
    """Final hidden state of one direction; embed (T, S, V)."""
    T, S, _ = embed.shape
    H = w_hh.shape[1]
    h = embed.new_zeros(S, H)
    c = embed.new_zeros(S, H)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gates = embed[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, g, o = gates.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
    return h


def clip_forward(sd: Dict[str, Tensor], appearance_clips: Tensor, train: bool = False,
                 masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """(B, N, T, V) -> (B, N, 2H).  masks: 'in' (B, N, T, V), 'out' (B*N, 2H) keep-masks (1 = keep) of the two nn.Dropout
    sites in call order; needed iff train."""
    masks = masks or {}
    bs, n_clips, n_frames, feat = appearance_clips.shape
    x = appearance_clips
    if train:
        x = x * masks["in"].view_as(x) / (1.0 - DROP_IN)                      # :214 embedding_dropout
    embed = torch.tanh(x)                                                      # :214
    embed = embed.permute(2, 0, 1, 3).reshape(n_frames, bs * n_clips, feat)    # :215-217
    hs = []
    for sfx, rev in (("", False), ("_reverse", True)):                        # :221 nn.LSTM(bidirectional=True)
        hs.append(_direction(embed, sd[PREFIX + "weight_ih_l0" + sfx], sd[PREFIX + "weight_hh_l0" + sfx],
                             sd[PREFIX + "bias_ih_l0" + sfx], sd[PREFIX + "bias_hh_l0" + sfx], rev))
    out = torch.cat(hs, dim=-1)                                                # :223
    if train:
        out = out * masks["out"].view_as(out) / (1.0 - DROP_OUT)              # :225 finalvisual_dropout
    return out.view(bs, n_clips, -1)                                           # :226
