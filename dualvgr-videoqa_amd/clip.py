"""The recurrent encoders in front of the unit on libdvgr_hip.so.

`BiLSTM` is a one-layer bidirectional LSTM over padded, optionally variable-length sequences with torch.nn.LSTM's parameter
names, shapes and registration order; `QuestionEncoder` uses two of them to restate `InputUnitLinguisticDynamic`
(model/Preprocessing.py:89-127, with `DynamicRNN` :7-45).  `ClipEncoder` holds the parameters of the reference's `VisualAppearanceEncoder` (model/Preprocessing.py:191-234) under
the same names and shapes (`encoder.weight_ih_l0`, ... `encoder.bias_hh_l0_reverse`), so that slice of a reference
checkpoint loads with strict=True, and computes dropout -> tanh -> BiLSTM over the frames of every clip -> final states
-> dropout through the C ABI of include/dvgr_clip.h -- forward and backward.  No CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L_
from .inputs import WordFeatures

CP_NAMES = ["weight_ih_l0", "weight_ih_l0_reverse", "weight_hh_l0", "weight_hh_l0_reverse", "bias_ih_l0",
            "bias_ih_l0_reverse", "bias_hh_l0", "bias_hh_l0_reverse"]          # enum dvgr_clip_param order
# registration order of torch.nn.LSTM (= key order of a reference state_dict)
LSTM_ORDER = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
              "bias_ih_l0_reverse", "bias_hh_l0_reverse"]


LSTM_EMBED_INPUT, LSTM_DROP_FINAL = 1, 2


class LstmIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "lens", "out_final", "out_seq")]


class LstmGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g_final", "g_seq", "g_x", "gparams")]


class ClipDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("S", "T", "V", "H", "dtype")]


CLIP_SYMBOLS = [
    ("dvgr_clip_check_dims", C.c_int, [C.POINTER(ClipDims)]),
    ("dvgr_clip_param_count", C.c_int64, [C.POINTER(ClipDims)]),
    ("dvgr_clip_param_offset", C.c_int, [C.POINTER(ClipDims), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64)]),
    ("dvgr_clip_saved_bytes", C.c_size_t, [C.POINTER(ClipDims)]),
    ("dvgr_clip_workspace_bytes", C.c_size_t, [C.POINTER(ClipDims)]),
    ("dvgr_clip_forward", C.c_int, [C.POINTER(ClipDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_size_t, C.c_int, C.c_uint64, C.c_void_p]),
    ("dvgr_clip_backward", C.c_int, [C.POINTER(ClipDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_size_t, C.c_int, C.c_uint64, C.c_void_p]),
    ("dvgr_lstm_forward", C.c_int, [C.POINTER(ClipDims), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_size_t, C.c_int, C.c_uint64, C.c_void_p]),
    ("dvgr_lstm_backward", C.c_int, [C.POINTER(ClipDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_void_p]),
    ("dvgr_clip_dbg_keep_mask", C.c_int, [C.c_uint64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
]
_bound = False


def load():
    global _bound
    lib = L_.load()
    if not _bound:
        for name, restype, argtypes in CLIP_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = True
    return lib


class _LstmFn(torch.autograd.Function):
    """(x (S, T, V), lens or None) -> (final (S, 2H), seq (S, T, 2H)); `want` selects which of the two are computed."""

    @staticmethod
    def forward(ctx, mod: "BiLSTM", x, lens, flags, want_final, want_seq, *params):
        lib = load()
        if not x.is_cuda:
            raise RuntimeError("BiLSTM runs on a ROCm device only (no CPU fallback)")
        x = x.contiguous().float()
        S, T, V = x.shape
        if V != mod.input_size:
            raise RuntimeError("inconsistent input shape for BiLSTM.forward")
        H = mod.hidden_size
        dims = ClipDims(S, T, V, H, mod.dtype_code)
        L_.check(lib.dvgr_clip_check_dims(C.byref(dims)), "dvgr_clip_check_dims")
        dev = x.device
        if lens is not None:
            lens = lens.to(device=dev, dtype=torch.int64).contiguous()
        flat = mod.flat_params()
        final = torch.empty((S, 2 * H), device=dev, dtype=torch.float32) if want_final else None
        seq = torch.empty((S, T, 2 * H), device=dev, dtype=torch.float32) if want_seq else None
        saved = torch.empty(lib.dvgr_clip_saved_bytes(C.byref(dims)), device=dev, dtype=torch.uint8)
        ws = mod.workspace(lib.dvgr_clip_workspace_bytes(C.byref(dims)), dev)
        train = bool(mod.training) and mod.dropout_enabled     # the only training-mode behaviour of the LSTM is dropout
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train and flags else 0
        if mod.forced_seed is not None:
            seed = int(mod.forced_seed)
        io = LstmIO(x.data_ptr(), lens.data_ptr() if lens is not None else None, final.data_ptr() if want_final else None,
                    seq.data_ptr() if want_seq else None)
        stream = torch.cuda.current_stream(dev).cuda_stream
        L_.check(lib.dvgr_lstm_forward(C.byref(dims), flat.data_ptr(), C.byref(io), int(flags), saved.data_ptr(), saved.numel(),
                                       ws.data_ptr(), ws.numel(), int(train), seed, stream), "dvgr_lstm_forward")
        ctx.mod, ctx.dims, ctx.train, ctx.seed, ctx.flags = mod, dims, train, seed, int(flags)
        ctx.lens = lens
        ctx.want = (want_final, want_seq)
        ctx.need_gx = ctx.needs_input_grad[1]
        ctx.save_for_backward(saved, flat)
        ctx.set_materialize_grads(False)
        mod.last_seed = seed
        outs = (final if want_final else x.new_zeros(0), seq if want_seq else x.new_zeros(0))
        ctx.mark_non_differentiable(*[o for o, w in zip(outs, ctx.want) if not w])
        return outs

    @staticmethod
    def backward(ctx, g_final, g_seq):
        lib = load()
        mod, dims = ctx.mod, ctx.dims
        saved, flat = ctx.saved_tensors
        dev = saved.device
        S, T, V, H = dims.S, dims.T, dims.V, dims.H
        g_final = g_final.contiguous().float() if (g_final is not None and ctx.want[0]) else None
        g_seq = g_seq.contiguous().float() if (g_seq is not None and ctx.want[1]) else None
        if g_final is None and g_seq is None:
            return (None,) * (6 + len(mod._plist))
        gflat = torch.empty_like(flat)
        g_x = torch.empty((S, T, V), device=dev, dtype=torch.float32) if ctx.need_gx else None
        ws = mod.workspace(lib.dvgr_clip_workspace_bytes(C.byref(dims)), dev)
        io = LstmIO(None, ctx.lens.data_ptr() if ctx.lens is not None else None, None, None)
        gr = LstmGrads(g_final.data_ptr() if g_final is not None else None, g_seq.data_ptr() if g_seq is not None else None,
                       g_x.data_ptr() if g_x is not None else None, gflat.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        L_.check(lib.dvgr_lstm_backward(C.byref(dims), flat.data_ptr(), C.byref(io), C.byref(gr), ctx.flags, saved.data_ptr(),
                                        saved.numel(), ws.data_ptr(), ws.numel(), int(ctx.train), ctx.seed, stream),
                 "dvgr_lstm_backward")
        if mod.grad_sink is not None:             # data parallel: the flat gradient buffer goes to the reducer right here
            mod.grad_sink(mod, gflat)             # (parallel.NativeParamReducer: one in-place all-reduce, enqueued behind this call)
        pgrads = [gflat[o:o + p.numel()].view(p.shape) for p, o in zip(mod._plist, mod._table)]
        return (None, g_x, None, None, None, None, *pgrads)


class BiLSTM(nn.Module):
    """One-layer bidirectional LSTM; parameters named, shaped and ordered as torch.nn.LSTM(bidirectional=True)'s, so the
    matching slice of a reference checkpoint loads with strict=True.  `run()` is the batch-first call."""

    def __init__(self, input_size: int, hidden_size: int, dtype: str = "fp32"):
        super().__init__()
        self.input_size, self.hidden_size = int(input_size), int(hidden_size)
        self.dtype_code = {"fp32": L_.DVGR_F32, "bf16": L_.DVGR_BF16, "bf16x3": L_.DVGR_BF16X3}[dtype]
        V, H = self.input_size, self.hidden_size
        k = 1.0 / (H ** 0.5)                      # torch.nn.LSTM.reset_parameters: U(-1/sqrt(H), 1/sqrt(H))
        for name in LSTM_ORDER:
            shape = (4 * H, V) if "weight_ih" in name else (4 * H, H) if "weight_hh" in name else (4 * H,)
            setattr(self, name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))
        self.forced_seed = None
        self.dropout_enabled = True   # False: training mode without the dropout sites (deterministic step comparisons)
        self.last_seed = 0
        self.grad_sink = None         # callable(module, flat_gradient) or None: see _LstmFn.backward
        self._plist = [getattr(self, n) for n in CP_NAMES]
        self._table = None
        self._flat = None
        self._ptrs = None
        self._ws = None
        self._build_table()

    def flatten_parameters(self):          # model/Preprocessing.py:118, :220 call it; the flat buffer is always contiguous
        return None

    def _build_table(self):
        dims = ClipDims(1, 1, self.input_size, self.hidden_size, self.dtype_code)
        try:
            lib = load()
        except RuntimeError:
            self._table = None           # library not built: the module can still hold / load parameters
            return
        tab = []
        for kind in range(len(CP_NAMES)):
            off, rows, cols = C.c_int64(), C.c_int64(), C.c_int64()
            L_.check(lib.dvgr_clip_param_offset(C.byref(dims), kind, C.byref(off), C.byref(rows), C.byref(cols)),
                     "dvgr_clip_param_offset")
            tab.append(off.value)
        self._table = tab
        self._total = int(lib.dvgr_clip_param_count(C.byref(dims)))

    def _flatten(self):
        dev = self._plist[0].device
        flat = torch.zeros(self._total, device=dev, dtype=torch.float32)
        for p, off in zip(self._plist, self._table):
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
        self._flat = flat
        self._ptrs = tuple(p.data_ptr() for p in self._plist)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if getattr(self, "_table", None) is not None:
            self._flatten()
        return out

    def flat_params(self) -> torch.Tensor:
        if self._table is None:
            self._build_table()
        if self._flat is None or tuple(p.data_ptr() for p in self._plist) != self._ptrs:
            self._flatten()
        return self._flat

    def workspace(self, nbytes: int, dev) -> torch.Tensor:
        """Scratch of the two calls (hundreds of MB at training shapes): kept between steps instead of reallocated."""
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        return self._ws

    def run(self, x, lens=None, flags: int = 0, want_final: bool = True, want_seq: bool = False):
        """x (S, T, V) -> (final (S, 2H) or None, seq (S, T, 2H) or None)."""
        final, seq = _LstmFn.apply(self, x, lens, int(flags), bool(want_final), bool(want_seq), *self._plist)
        return (final if want_final else None), (seq if want_seq else None)


class ClipEncoder(nn.Module):
    """Drop-in for the reference's VisualAppearanceEncoder(appearance_dim, module_dim): (B, N, frames, feat) -> (B, N, D)."""

    def __init__(self, appearance_dim: int = 2048, module_dim: int = 512, dtype: str = "fp32"):
        super().__init__()
        self.appearance_dim, self.module_dim = int(appearance_dim), int(module_dim)
        self.encoder = BiLSTM(self.appearance_dim, self.module_dim // 2, dtype=dtype)

    @property
    def forced_seed(self):
        return self.encoder.forced_seed

    @forced_seed.setter
    def forced_seed(self, v):
        self.encoder.forced_seed = v

    def forward(self, appearance_clips):
        bs, n_clips, n_frames, feat = appearance_clips.shape
        if feat != self.appearance_dim:
            raise RuntimeError("inconsistent input shape for ClipEncoder.forward")
        x = appearance_clips.reshape(bs * n_clips, n_frames, feat)
        final, _ = self.encoder.run(x, None, LSTM_EMBED_INPUT | LSTM_DROP_FINAL)
        return final.view(bs, n_clips, self.module_dim)


class _PackedRNN(nn.Module):
    """`DynamicRNN` (model/Preprocessing.py:7-45): the per-step outputs of a packed BiLSTM, zero past each length."""

    def __init__(self, input_size, hidden_size, dtype):
        super().__init__()
        self.rnn = BiLSTM(input_size, hidden_size, dtype=dtype)

    def forward(self, x, seq_len, max_len):
        _, seq = self.rnn.run(x, seq_len, 0, want_final=False, want_seq=True)
        return seq


class QuestionEncoder(nn.Module):
    """Drop-in for the reference's InputUnitLinguisticDynamic (model/Preprocessing.py:89-127): embedding -> dropout -> tanh =
    word features (one gather kernel; backward: one scatter-add kernel into the table's gradient, include/dvgr_inputs.h);
    one BiLSTM gives the per-word dynamic embedding (zero past the question length), a second one the question embedding
    (final states -> dropout 0.18).  `encoder_embed` is kept as the parameter holder (state_dict key, U(-1, 1) init)."""

    def __init__(self, vocab_size, wordvec_dim=300, rnn_dim=512, dtype: str = "fp32"):
        super().__init__()
        self.encoder_embed = nn.Embedding(vocab_size, wordvec_dim)
        self.concatRNN = _PackedRNN(wordvec_dim, rnn_dim // 2, dtype)
        self.encoder = BiLSTM(wordvec_dim, rnn_dim // 2, dtype=dtype)
        self.embedding_dropout = nn.Dropout(p=0.15)          # holds p only: the kernel's site drops with p = 0.15 or not at all
        self.word_features = WordFeatures()

    def forward(self, questions, question_len):
        max_len = questions.size(1)
        # model/Preprocessing.py:112-113 as one gather kernel (and one scatter-add kernel backward), include/dvgr_inputs.h
        if self.embedding_dropout.p not in (0.0, 0.15):
            raise ValueError("the word-feature kernel drops with the reference's p = 0.15 (or not at all)")
        words = self.word_features(self.encoder_embed.weight, questions, self.training and self.embedding_dropout.p > 0.0)
        output_embedding = self.concatRNN(words, question_len, max_len)
        question_embedding, _ = self.encoder.run(words, question_len, LSTM_DROP_FINAL)
        return question_embedding, words, output_embedding
