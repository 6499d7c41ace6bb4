"""Whole-model wrapper: `DualVGR` with the reference's constructor, `forward()` and
state_dict surface (model/models.py:34-83).

The reasoning unit, the answer head and the appearance encoder (the BiLSTM over the frames of every clip, 90 % of the
encoders' time) are this package's HIP paths (`DualVGRUnit_multiple`, `AnswerHead`, `ClipEncoder`).  The small encoders
left -- question embedding + two BiLSTMs over <= T words of B questions, the motion Linear (model/Preprocessing.py:7-45,
:89-127; model/models.py:46) -- are restated here on stock PyTorch-ROCm modules (MIOpen LSTM, hipBLASLt Linear), as
SURVEY.md ranks them: "start with MIOpen via nn.LSTM".  `appearance_encoder="miopen"` puts the appearance encoder on
nn.LSTM as well (the first stage of this row, kept for A/B timing).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .clip import ClipEncoder, QuestionEncoder
from .head import AnswerHead
from .unit import DualVGRUnit_multiple


class _DynamicRNN(nn.Module):
    """Packed (length-aware) batch-first RNN; padded back to `max_len` (model/Preprocessing.py:7-45)."""

    def __init__(self, input_size, hidden_size, bidirectional=True, rnn_encoder="LSTM"):
        super().__init__()
        self.rnn = getattr(nn, rnn_encoder)(input_size, hidden_size, num_layers=1, bias=True, batch_first=True,
                                            bidirectional=bidirectional)

    def forward(self, x, seq_len, max_len):
        packed = nn.utils.rnn.pack_padded_sequence(x, seq_len.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = self.rnn(packed)
        out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
        if out.shape[1] < max_len:
            out = F.pad(out, [0, 0, 0, max_len - out.shape[1]])
        return out


class InputUnitLinguisticDynamic(nn.Module):
    """Question encoder (model/Preprocessing.py:89-127): embedding -> dropout -> tanh = word features; one BiLSTM gives the
    per-word dynamic embedding (zero past the question length), a second one the question embedding (final states)."""

    def __init__(self, vocab_size, wordvec_dim=300, rnn_dim=512, textual_encoder="LSTM"):
        super().__init__()
        self.encoder_embed = nn.Embedding(vocab_size, wordvec_dim)
        self.concatRNN = _DynamicRNN(wordvec_dim, rnn_dim // 2, bidirectional=True, rnn_encoder=textual_encoder)
        self.encoder = getattr(nn, textual_encoder)(wordvec_dim, rnn_dim // 2, batch_first=True, bidirectional=True)
        self.embedding_dropout = nn.Dropout(p=0.15)
        self.final_dropout = nn.Dropout(0.18)

    def forward(self, questions, question_len):
        max_len = questions.size(1)
        words = torch.tanh(self.embedding_dropout(self.encoder_embed(questions)))
        output_embedding = self.concatRNN(words, question_len, max_len)
        packed = nn.utils.rnn.pack_padded_sequence(words, question_len.cpu(), batch_first=True, enforce_sorted=False)
        _, (h, _) = self.encoder(packed)
        question_embedding = self.final_dropout(torch.cat([h[0], h[1]], -1))
        return question_embedding, words, output_embedding


class VisualAppearanceEncoder(nn.Module):
    """Appearance encoder (model/Preprocessing.py:191-234): dropout -> tanh, a BiLSTM over the frames of every clip
    (sequence-first, batch = B * N clips), concatenated final states -> dropout."""

    def __init__(self, appearance_dim=2048, module_dim=512):
        super().__init__()
        self.encoder = nn.LSTM(appearance_dim, module_dim // 2, batch_first=False, bidirectional=True)
        self.embedding_dropout = nn.Dropout(p=0.15)
        self.finalvisual_dropout = nn.Dropout(p=0.18)
        self.module_dim = module_dim

    def forward(self, appearance_clips):
        bs, n_clips, n_frames, feat = appearance_clips.shape
        embed = torch.tanh(self.embedding_dropout(appearance_clips))
        embed = embed.permute(2, 0, 1, 3).reshape(n_frames, bs * n_clips, feat)     # (frames, B * N, feat)
        _, (h, _) = self.encoder(embed)
        out = self.finalvisual_dropout(torch.cat([h[0], h[1]], -1))
        return out.view(bs, n_clips, self.module_dim)


class DualVGR(nn.Module):
    """model/models.py:34-83 with the unit and the answer head on HIP.  Same constructor arguments, same forward
    signature and 7-tuple, same state_dict keys (a reference checkpoint loads with strict=True)."""

    def __init__(self, vision_dim=2048, module_dim=768, word_dim=300, vocab=None, num_of_nodes=8, graph_module="GAT",
                 graph_layers=1, unit_layers=2, dtype="fp32", appearance_encoder="hip",
                 question_encoder="hip"):
        super().__init__()
        n_q = len(vocab["question_token_to_idx"])
        self.num_classes = len(vocab["answer_token_to_idx"])
        if question_encoder == "hip":
            self.linguistic_input_unit = QuestionEncoder(vocab_size=n_q, wordvec_dim=word_dim, rnn_dim=module_dim, dtype=dtype)
        elif question_encoder == "miopen":
            self.linguistic_input_unit = InputUnitLinguisticDynamic(vocab_size=n_q, wordvec_dim=word_dim, rnn_dim=module_dim)
        else:
            raise ValueError("question_encoder must be 'hip' or 'miopen'")
        if appearance_encoder == "hip":
            self.visual_appearance_input_unit = ClipEncoder(appearance_dim=vision_dim, module_dim=module_dim, dtype=dtype)
        elif appearance_encoder == "miopen":
            self.visual_appearance_input_unit = VisualAppearanceEncoder(appearance_dim=vision_dim, module_dim=module_dim)
        else:
            raise ValueError("appearance_encoder must be 'hip' or 'miopen'")
        self.visual_motion_input_unit = nn.Linear(vision_dim, module_dim)
        self.visual_input_unit = DualVGRUnit_multiple(word_dim=word_dim, module_dim=module_dim, num_of_nodes=num_of_nodes,
                                                      appearance_graph_layers=graph_layers, motion_graph_layers=graph_layers,
                                                      graph_module=graph_module, unit_layers=unit_layers, dtype=dtype)
        head = AnswerHead(module_dim=module_dim, num_answers=self.num_classes, dtype=dtype)
        # the reference keeps the two halves of the head as top-level modules; register them under those names and keep
        # the AnswerHead object itself out of the module tree (no duplicate state_dict keys)
        self.feature_aggregation = head.feature_aggregation
        self.output_unit = head.output_unit
        object.__setattr__(self, "_head", head)
        self.overlap_encoders = True
        for m in self.modules():                    # model/utils.py:8-33 init_modules(w_init="xavier_uniform")
            if isinstance(m, (nn.Linear, nn.LSTM)) or type(m).__name__ == "BiLSTM":
                for name, p in m.named_parameters(recurse=False):
                    if "bias" in name:
                        nn.init.zeros_(p)
                    elif p.dim() >= 2:
                        nn.init.xavier_uniform_(p)
        nn.init.uniform_(self.linguistic_input_unit.encoder_embed.weight, -1.0, 1.0)

    def _encoder_stream(self, dev):
        return _lib.streams(dev)["encoder"]

    def train(self, mode: bool = True):
        super().train(mode)
        self._head.train(mode)
        return self

    def forward(self, video_appearance_feat, video_motion_feat, question, question_len):
        # The question encoder (chains of small launches over <= T words of B questions) and the visual encoders are
        # independent: run the former on a second stream so its kernels fill the gaps of the latter.  autograd replays
        # each node's backward on the stream of its forward, so the backward overlaps the same way.
        dev = video_appearance_feat.device
        if dev.type == "cuda" and self.overlap_encoders:
            cur = torch.cuda.current_stream(dev)
            side = self._encoder_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                question_embedding, word_embedding, dynamic_question_embedding = self.linguistic_input_unit(question, question_len)
            app = self.visual_appearance_input_unit(video_appearance_feat)
            mot = self.visual_motion_input_unit(video_motion_feat)
            cur.wait_stream(side)
            for t in (question_embedding, word_embedding, dynamic_question_embedding):
                t.record_stream(cur)
        else:
            question_embedding, word_embedding, dynamic_question_embedding = self.linguistic_input_unit(question, question_len)
            app = self.visual_appearance_input_unit(video_appearance_feat)
            mot = self.visual_motion_input_unit(video_motion_feat)
        visual_embedding, aq_embed, mq_embed, com_app, com_motion, aq_fusion, mq_fusion = self.visual_input_unit(
            app, mot, dynamic_question_embedding, word_embedding, question_len)
        out = self._head(question_embedding, visual_embedding)
        return out, aq_embed, mq_embed, com_app, com_motion, aq_fusion, mq_fusion
