"""Whole-model wrapper: `DualVGR` with the reference's constructor, `forward()` and
state_dict surface (model/models.py:34-83).

The reasoning unit, the answer head, the appearance encoder (the BiLSTM over the frames of every clip) and the question
encoder (two packed BiLSTMs over <= T words of B questions) are this package's HIP paths (`DualVGRUnit_multiple`,
`AnswerHead`, `ClipEncoder`, `QuestionEncoder`).  There is ONE backend: the stock-PyTorch (MIOpen) restatements of the
encoders that round 1 kept here for A/B timing live in tools/stock_encoders.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .clip import ClipEncoder, QuestionEncoder
from .head import AnswerHead
from .inputs import HipLinear
from .unit import DualVGRUnit_multiple


def init_modules(model: nn.Module) -> None:
    """model/utils.py:8-33 as invoked by model/models.py:52 (`init_modules(self.modules(), w_init="xavier_uniform")`):
    xavier-uniform (gain 1) on the weight of every Linear and on every LSTM `weight_*`, zeros on their biases.  In the
    reference this runs over EVERY nn.Linear of the model, i.e. also over the unit's (QueryAttn, QueryPunish, the GAT
    heads' W and a, the SF-attention projections, the MFB linears -- overriding the GAT's gain-1.414 and the MFB's
    N(0, 0.01) constructors) and the answer head's.  Here those layers are parameter holders (`_Holder`: the arithmetic
    runs in the HIP kernels), so the loop matches on them as well.  BatchNorm and the embedding are left alone, as there
    (the embedding is then drawn from U(-1, 1), model/models.py:53)."""
    from .clip import BiLSTM
    from .unit import _Holder
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Linear, _Holder)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.LSTM, nn.GRU, BiLSTM)):
                for name, p in m.named_parameters(recurse=False):
                    if "bias" in name:
                        nn.init.zeros_(p)
                    elif "weight" in name:
                        nn.init.xavier_uniform_(p)


class DualVGR(nn.Module):
    """model/models.py:34-83 with the unit and the answer head on HIP.  Same constructor arguments, same forward
    signature and 7-tuple, same state_dict keys (a reference checkpoint loads with strict=True)."""

    def __init__(self, vision_dim=2048, module_dim=768, word_dim=300, vocab=None, num_of_nodes=8, graph_module="GAT",
                 graph_layers=1, unit_layers=2, dtype="fp32"):
        super().__init__()
        n_q = len(vocab["question_token_to_idx"])
        self.num_classes = len(vocab["answer_token_to_idx"])
        self.linguistic_input_unit = QuestionEncoder(vocab_size=n_q, wordvec_dim=word_dim, rnn_dim=module_dim, dtype=dtype)
        self.visual_appearance_input_unit = ClipEncoder(appearance_dim=vision_dim, module_dim=module_dim, dtype=dtype)
        self.visual_motion_input_unit = HipLinear(vision_dim, module_dim, dtype=dtype)      # model/models.py:46 on the contraction engine
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
        self.last_raw = None
        init_modules(self)
        nn.init.uniform_(self.linguistic_input_unit.encoder_embed.weight, -1.0, 1.0)

    def set_dropout_enabled(self, flag: bool):
        """Training mode with every dropout site switched off (BatchNorm keeps using batch statistics): what a
        deterministic comparison with the reference's training step needs (tests/golden/make_golden_step.py)."""
        for m in self.modules():
            if hasattr(m, "dropout_enabled"):
                m.dropout_enabled = bool(flag)
        self._head.dropout_enabled = bool(flag)
        self.linguistic_input_unit.embedding_dropout.p = 0.15 if flag else 0.0
        return self

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
        raw = self.visual_input_unit.forward_raw(app, mot, dynamic_question_embedding, word_embedding, question_len)
        self.last_raw = raw                      # the four lists as the (L, B, N, D) tensors the kernels wrote (trainer.TrainStep)
        visual_embedding, aq_embed, mq_embed = raw[0], raw[1], raw[2]
        com_app, com_motion, aq_fusion, mq_fusion = (list(t.unbind(0)) for t in raw[3:])
        out = self._head(question_embedding, visual_embedding)
        return out, aq_embed, mq_embed, com_app, com_motion, aq_fusion, mq_fusion
