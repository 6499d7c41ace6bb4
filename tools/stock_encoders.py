"""A/B timing aid, NOT product code: the reference's three encoders restated on stock PyTorch-ROCm modules (nn.LSTM ->
MIOpen, nn.Linear -> hipBLASLt) with the reference's parameter names, and `use_stock_encoders()` to swap them into a
`dualvgr_videoqa_amd.DualVGR` in place of the package's HIP encoders (bench.py --appearance-encoder / --question-encoder
miopen, tools/clip_time.py).  Reference: model/Preprocessing.py:7-45 (DynamicRNN), :89-127 (InputUnitLinguisticDynamic),
:191-234 (VisualAppearanceEncoder)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


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


def use_stock_encoders(model, appearance: bool = False, question: bool = False):
    """Replaces the HIP encoders of a DualVGR by the stock modules above, carrying the weights over (same state_dict keys).
    Returns the model."""
    dev = next(model.parameters()).device
    D = model.visual_input_unit.module_dim
    if appearance:
        old = model.visual_appearance_input_unit
        new = VisualAppearanceEncoder(appearance_dim=old.encoder.input_size, module_dim=D).to(dev)
        new.load_state_dict(old.state_dict(), strict=True)
        model.visual_appearance_input_unit = new
    if question:
        old = model.linguistic_input_unit
        new = InputUnitLinguisticDynamic(vocab_size=old.encoder_embed.weight.shape[0],
                                         wordvec_dim=old.encoder_embed.weight.shape[1], rnn_dim=D).to(dev)
        new.load_state_dict(old.state_dict(), strict=True)
        model.linguistic_input_unit = new
    model.train(model.training)
    return model
