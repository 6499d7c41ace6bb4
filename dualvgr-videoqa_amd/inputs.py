"""Host side of include/dvgr_inputs.h: the two input-side operators of the reference model
that are neither an LSTM nor the unit --

  * `word_features(table, questions, train)`: tanh(dropout_0.15(embedding(questions))) of
    InputUnitLinguisticDynamic.forward (model/Preprocessing.py:112-113) as one gather kernel, and its backward as one
    scatter-add kernel into the embedding table's gradient;
  * `HipLinear`: DualVGR.visual_motion_input_unit = nn.Linear(vision_dim, module_dim) (model/models.py:46, :74) on the
    unit's contraction engine.  A subclass of nn.Linear: same parameters, state_dict keys and default init.

No CPU fallback: both raise off a ROCm device."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L_

INPUT_SYMBOLS = [
    ("dvgr_words_forward", C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    ("dvgr_words_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint64,
                                      C.c_void_p, C.c_void_p]),
    ("dvgr_words_dbg_keep_mask", C.c_int, [C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
    ("dvgr_linear_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    ("dvgr_linear_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
]
_bound = False
DTYPES = {"fp32": L_.DVGR_F32, "bf16": L_.DVGR_BF16, "bf16x3": L_.DVGR_BF16X3}


def load():
    global _bound
    lib = L_.load()
    if not _bound:
        for name, restype, argtypes in INPUT_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = True
    return lib


def _need_device(t: torch.Tensor, what: str):
    if t.device.type != "cuda":
        raise RuntimeError(f"{what} runs on a ROCm device only (no CPU fallback); got a tensor on {t.device}")


class _WordFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, tokens, train, seed, bad):
        lib = load()
        _need_device(table, "word_features")
        tokens = tokens.contiguous()
        if tokens.dtype != torch.int64:
            tokens = tokens.long()
        table_c = table.contiguous()
        vocab, wd = table_c.shape
        words = torch.empty(*tokens.shape, wd, device=table.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(table.device).cuda_stream
        L_.check(lib.dvgr_words_forward(table_c.data_ptr(), vocab, wd, tokens.data_ptr(), tokens.numel(), int(train), int(seed),
                                        words.data_ptr(), bad.data_ptr() if bad is not None else None, stream), "dvgr_words_forward")
        ctx.save_for_backward(words, tokens)
        ctx.meta = (vocab, wd, int(train), int(seed))
        return words

    @staticmethod
    def backward(ctx, g):
        lib = load()
        words, tokens = ctx.saved_tensors
        vocab, wd, train, seed = ctx.meta
        g = g.contiguous()
        g_table = torch.empty(vocab, wd, device=g.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        L_.check(lib.dvgr_words_backward(g.data_ptr(), words.data_ptr(), tokens.data_ptr(), tokens.numel(), vocab, wd, train, seed,
                                         g_table.data_ptr(), stream), "dvgr_words_backward")
        return g_table, None, None, None, None


class WordFeatures:
    """Stateless helper the question encoder holds: draws the dropout seed, keeps the bad-token flag of the last call."""

    def __init__(self):
        self.forced_seed = None          # tests: fix the dropout seed
        self.last_seed = 0
        self._bad = None

    def __call__(self, table: torch.Tensor, questions: torch.Tensor, train: bool) -> torch.Tensor:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train else 0
        if self.forced_seed is not None:
            seed = int(self.forced_seed)
        self.last_seed = seed
        if self._bad is None or self._bad.device != table.device:
            self._bad = torch.zeros(1, device=table.device, dtype=torch.int32)
        return _WordFeatures.apply(table, questions, bool(train), seed, self._bad)

    def check_tokens(self):
        """Synchronises: raises if any call so far met a token outside the table (torch raises at the lookup itself)."""
        if self._bad is not None and int(self._bad.item()) != 0:
            self._bad.zero_()
            raise IndexError("word_features: a question token is outside the embedding table")


def words_keep_mask(seed: int, shape, device) -> torch.Tensor:
    """0/1 keep flags of the word-feature dropout site for `seed` (tests inject them into the oracle)."""
    lib = load()
    n = 1
    for s in shape:
        n *= int(s)
    keep = torch.empty(n, device=device, dtype=torch.float32)
    L_.check(lib.dvgr_words_dbg_keep_mask(int(seed), n, keep.data_ptr(), torch.cuda.current_stream(device).cuda_stream),
             "dvgr_words_dbg_keep_mask")
    return keep.view(*shape)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, dtype_code):
        lib = load()
        _need_device(x, "HipLinear")
        K = x.shape[-1]
        x2 = x.reshape(-1, K).contiguous()
        w_c = w.contiguous()
        N = w_c.shape[0]
        y = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        L_.check(lib.dvgr_linear_forward(x2.data_ptr(), w_c.data_ptr(), b.contiguous().data_ptr() if b is not None else None,
                                         y.data_ptr(), x2.shape[0], N, K, int(dtype_code), stream), "dvgr_linear_forward")
        ctx.save_for_backward(x2, w_c)
        ctx.meta = (tuple(x.shape), b is not None, int(dtype_code))
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        lib = load()
        x2, w = ctx.saved_tensors
        xshape, has_b, code = ctx.meta
        M, K = x2.shape
        N = w.shape[0]
        gy2 = gy.reshape(M, N).contiguous()
        gw = torch.empty_like(w)
        gb = torch.empty(N, device=gy.device, dtype=torch.float32) if has_b else None
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        stream = torch.cuda.current_stream(gy.device).cuda_stream
        L_.check(lib.dvgr_linear_backward(x2.data_ptr(), w.data_ptr(), gy2.data_ptr(), gx.data_ptr() if gx is not None else None,
                                          gw.data_ptr(), gb.data_ptr() if gb is not None else None, M, N, K, code, stream),
                 "dvgr_linear_backward")
        return (gx.view(*xshape) if gx is not None else None), gw, gb, None


class HipLinear(nn.Linear):
    """nn.Linear(in_features, out_features) whose forward / backward run on the unit's contraction engine
    (dvgr_linear_forward / dvgr_linear_backward).  `dtype`: 'fp32' | 'bf16x3' | 'bf16' as for the unit."""

    def __init__(self, in_features, out_features, bias=True, dtype: str = "fp32"):
        super().__init__(in_features, out_features, bias=bias)
        self.dtype_code = DTYPES[dtype]

    def forward(self, x):
        return _Linear.apply(x, self.weight, self.bias, self.dtype_code)
