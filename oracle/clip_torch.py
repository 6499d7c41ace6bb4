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


# ---- question encoder: InputUnitLinguisticDynamic (model/Preprocessing.py:89-127) with DynamicRNN (:7-45) ----------------
def lstm_shapes(prefix: str, input_size: int, hidden: int) -> Dict[str, tuple]:
    out = {}
    for sfx in ("", "_reverse"):
        out[prefix + "weight_ih_l0" + sfx] = (4 * hidden, input_size)
        out[prefix + "weight_hh_l0" + sfx] = (4 * hidden, hidden)
        out[prefix + "bias_ih_l0" + sfx] = (4 * hidden,)
        out[prefix + "bias_hh_l0" + sfx] = (4 * hidden,)
    return out


def question_param_shapes(vocab_size: int, wordvec_dim: int, rnn_dim: int) -> Dict[str, tuple]:
    out = {"encoder_embed.weight": (vocab_size, wordvec_dim)}
    out.update(lstm_shapes("concatRNN.rnn.", wordvec_dim, rnn_dim // 2))
    out.update(lstm_shapes("encoder.", wordvec_dim, rnn_dim // 2))
    return out


def packed_bilstm(sd: Dict[str, Tensor], prefix: str, x: Tensor, lens: Optional[Tensor]):
    """What pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence computes, without packing: sequence s
    takes part in step t iff t < lens[s]; otherwise its state passes through and the step's output is zero.
    x (S, T, V) batch-first -> (final (S, 2H) = [h_fwd(len-1) | h_rev(0)], seq (S, T, 2H))."""
    S, T, _ = x.shape
    finals, seqs = [], []
    for sfx, rev in (("", False), ("_reverse", True)):
        w_ih, w_hh = sd[prefix + "weight_ih_l0" + sfx], sd[prefix + "weight_hh_l0" + sfx]
        b_ih, b_hh = sd[prefix + "bias_ih_l0" + sfx], sd[prefix + "bias_hh_l0" + sfx]
        H = w_hh.shape[1]
        h = x.new_zeros(S, H)
        c = x.new_zeros(S, H)
        steps = [None] * T
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            gates = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
            i, f, g, o = gates.split(H, dim=1)
            c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h_new = torch.sigmoid(o) * torch.tanh(c_new)
            if lens is None:
                c, h = c_new, h_new
                steps[t] = h_new
            else:
                act = (t < lens).to(x.dtype).unsqueeze(1)
                c = act * c_new + (1.0 - act) * c
                h = act * h_new + (1.0 - act) * h
                steps[t] = act * h_new
        finals.append(h)
        seqs.append(torch.stack(steps, dim=1))
    return torch.cat(finals, dim=-1), torch.cat(seqs, dim=-1)


def question_forward(sd: Dict[str, Tensor], questions: Tensor, question_len: Tensor, train: bool = False,
                     masks: Optional[Dict[str, Tensor]] = None):
    """-> (question_embedding (B, D), words (B, T, Wd), output_embedding (B, T, D)).
    masks: 'emb' (B, T, Wd) p = 0.15, 'final' (B, D) p = 0.18 keep-masks; needed iff train."""
    masks = masks or {}
    emb = sd["encoder_embed.weight"][questions]                                # :112
    if train:
        emb = emb * masks["emb"] / (1.0 - DROP_IN)                             # :113 embedding_dropout(0.15)
    words = torch.tanh(emb)                                                    # :113
    _, output_embedding = packed_bilstm(sd, "concatRNN.rnn.", words, question_len)   # :114 -> DynamicRNN :17-45
    question_embedding, _ = packed_bilstm(sd, "encoder.", words, question_len)       # :116-121
    if train:
        question_embedding = question_embedding * masks["final"] / (1.0 - DROP_OUT)  # :122 final_dropout(0.18)
    return question_embedding, words, output_embedding
