"""The appearance encoder in front of the unit on libdvgr_hip.so.

`ClipEncoder` holds the parameters of the reference's `VisualAppearanceEncoder` (model/Preprocessing.py:191-234) under
the same names and shapes (`encoder.weight_ih_l0`, ... `encoder.bias_hh_l0_reverse`), so that slice of a reference
checkpoint loads with strict=True, and computes dropout -> tanh -> BiLSTM over the frames of every clip -> final states
-> dropout through the C ABI of include/dvgr_clip.h -- forward and backward.  No CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L_

CP_NAMES = ["weight_ih_l0", "weight_ih_l0_reverse", "weight_hh_l0", "weight_hh_l0_reverse", "bias_ih_l0",
            "bias_ih_l0_reverse", "bias_hh_l0", "bias_hh_l0_reverse"]          # enum dvgr_clip_param order
# registration order of torch.nn.LSTM (= key order of a reference state_dict)
LSTM_ORDER = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
              "bias_ih_l0_reverse", "bias_hh_l0_reverse"]


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


class _LSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM(bidirectional=True)'s names, shapes and registration order."""

    def __init__(self, V, H):
        super().__init__()
        for name in LSTM_ORDER:
            shape = (4 * H, V) if "weight_ih" in name else (4 * H, H) if "weight_hh" in name else (4 * H,)
            setattr(self, name, nn.Parameter(torch.zeros(shape)))

    def flatten_parameters(self):          # model/Preprocessing.py:220 calls it; the flat buffer is always contiguous
        return None


class _ClipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "ClipEncoder", x, *params):
        lib = load()
        if not x.is_cuda:
            raise RuntimeError("ClipEncoder runs on a ROCm device only (no CPU fallback)")
        x = x.contiguous().float()
        bs, n_clips, n_frames, feat = x.shape
        if feat != mod.appearance_dim:
            raise RuntimeError("inconsistent input shape for ClipEncoder.forward")
        dims = ClipDims(bs * n_clips, n_frames, feat, mod.module_dim // 2, mod.dtype_code)
        L_.check(lib.dvgr_clip_check_dims(C.byref(dims)), "dvgr_clip_check_dims")
        dev = x.device
        flat = mod.flat_params()
        out = torch.empty((bs, n_clips, mod.module_dim), device=dev, dtype=torch.float32)
        saved = torch.empty(lib.dvgr_clip_saved_bytes(C.byref(dims)), device=dev, dtype=torch.uint8)
        ws = mod.workspace(lib.dvgr_clip_workspace_bytes(C.byref(dims)), dev)
        train = bool(mod.training)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train else 0
        if mod.forced_seed is not None:
            seed = int(mod.forced_seed)
        stream = torch.cuda.current_stream(dev).cuda_stream
        L_.check(lib.dvgr_clip_forward(C.byref(dims), flat.data_ptr(), x.data_ptr(), out.data_ptr(), saved.data_ptr(), saved.numel(),
                                       ws.data_ptr(), ws.numel(), int(train), seed, stream), "dvgr_clip_forward")
        ctx.mod, ctx.dims, ctx.train, ctx.seed = mod, dims, train, seed
        ctx.save_for_backward(saved, flat)
        mod.last_seed = seed
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = load()
        mod, dims = ctx.mod, ctx.dims
        saved, flat = ctx.saved_tensors
        dev = saved.device
        g = g_out.contiguous().float()
        gflat = torch.empty_like(flat)
        ws = mod.workspace(lib.dvgr_clip_workspace_bytes(C.byref(dims)), dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        L_.check(lib.dvgr_clip_backward(C.byref(dims), flat.data_ptr(), g.data_ptr(), gflat.data_ptr(), saved.data_ptr(),
                                        saved.numel(), ws.data_ptr(), ws.numel(), int(ctx.train), ctx.seed, stream),
                 "dvgr_clip_backward")
        pgrads = [gflat[o:o + p.numel()].view(p.shape) for p, o in zip(mod._plist, mod._table)]
        return (None, None, *pgrads)


class ClipEncoder(nn.Module):
    """Drop-in for the reference's VisualAppearanceEncoder(appearance_dim, module_dim): (B, N, frames, feat) -> (B, N, D)."""

    def __init__(self, appearance_dim: int = 2048, module_dim: int = 512, dtype: str = "fp32"):
        super().__init__()
        self.appearance_dim, self.module_dim = int(appearance_dim), int(module_dim)
        self.dtype_code = {"fp32": L_.DVGR_F32, "bf16": L_.DVGR_BF16, "bf16x3": L_.DVGR_BF16X3}[dtype]
        H = self.module_dim // 2
        self.encoder = _LSTMParams(self.appearance_dim, H)
        self.forced_seed = None
        self.last_seed = 0
        self._plist = [getattr(self.encoder, n) for n in CP_NAMES]
        self._table = None
        self._flat = None
        self._ptrs = None
        self._ws = None
        k = 1.0 / (H ** 0.5)                      # torch.nn.LSTM.reset_parameters: U(-1/sqrt(H), 1/sqrt(H))
        with torch.no_grad():
            for p in self._plist:
                p.uniform_(-k, k)
        self._build_table()

    def _build_table(self):
        dims = ClipDims(1, 1, self.appearance_dim, self.module_dim // 2, self.dtype_code)
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

    def forward(self, appearance_clips):
        return _ClipFn.apply(self, appearance_clips, *self._plist)
