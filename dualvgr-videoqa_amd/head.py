"""The step after the unit: clip read-out + classifier on libdvgr_hip.so.

`AnswerHead` holds the parameters of the reference's `DualVGR.feature_aggregation` (`ContextSelfAttn`,
model/AnswerDecoder.py:155-182) and `DualVGR.output_unit` (`SimpleOutputUnitOpenEnded`, :184-202) under the same names and
shapes, so the corresponding slice of a reference checkpoint loads with strict=True, and computes
`output_unit(question_embedding, feature_aggregation(visual_feature))` (model/models.py:80-81) through the C ABI of
include/dvgr_head.h -- forward and backward.  No CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L_
from .unit import _Holder

(HP_VPROJ_W, HP_ATTN_W, HP_ATTN_B, HP_QPROJ_W, HP_QPROJ_B, HP_FC1_W, HP_FC1_B, HP_BN_W, HP_BN_B, HP_FC2_W, HP_FC2_B,
 HP_KINDS) = range(12)


HEAD_TRAIN_NO_DROPOUT = 2      # include/dvgr_head.h


class HeadDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "D", "C", "dtype")]


class HeadIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("vis", "qe", "logits", "run_mean", "run_var")]


class HeadGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g_logits", "g_vis", "g_qe", "gparams")]


HEAD_SYMBOLS = [
    ("dvgr_head_check_dims", C.c_int, [C.POINTER(HeadDims)]),
    ("dvgr_head_param_count", C.c_int64, [C.POINTER(HeadDims)]),
    ("dvgr_head_param_offset", C.c_int, [C.POINTER(HeadDims), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64)]),
    ("dvgr_head_saved_bytes", C.c_size_t, [C.POINTER(HeadDims)]),
    ("dvgr_head_workspace_bytes", C.c_size_t, [C.POINTER(HeadDims)]),
    ("dvgr_head_forward", C.c_int, [C.POINTER(HeadDims), C.c_void_p, C.POINTER(HeadIO), C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_size_t, C.c_int, C.c_uint64, C.c_void_p]),
    ("dvgr_head_backward", C.c_int, [C.POINTER(HeadDims), C.c_void_p, C.POINTER(HeadIO), C.POINTER(HeadGrads), C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_void_p]),
    ("dvgr_head_cross_entropy", C.c_int, [C.POINTER(HeadDims), C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    ("dvgr_head_dbg_keep_mask", C.c_int, [C.c_uint64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
]
_bound = False


def load():
    global _bound
    lib = L_.load()
    if not _bound:
        for name, restype, argtypes in HEAD_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = True
    return lib


class _BNHolder(nn.Module):
    """Parameter / buffer container with nn.BatchNorm1d's names."""

    def __init__(self, D):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(D))
        self.bias = nn.Parameter(torch.zeros(D))
        self.register_buffer("running_mean", torch.zeros(D))
        self.register_buffer("running_var", torch.ones(D))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Readout(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.v_proj = _Holder(D, D, bias=False)
        self.attn = _Holder(1, D)


class _Classifier(nn.Module):
    def __init__(self, D, Cn):
        super().__init__()
        self.question_proj = _Holder(D, D)
        # indices as in nn.Sequential(Dropout, Linear, ELU, BatchNorm1d, Dropout, Linear) of the reference
        self.classifier = nn.Sequential(nn.Identity(), _Holder(D, 2 * D), nn.Identity(), _BNHolder(D), nn.Identity(),
                                        _Holder(Cn, D))


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "AnswerHead", qe, vis, *params):
        lib = load()
        if not vis.is_cuda:
            raise RuntimeError("AnswerHead runs on a ROCm device only (no CPU fallback)")
        vis, qe = vis.contiguous().float(), qe.contiguous().float()
        B, N, D = vis.shape
        if qe.shape != (B, D) or D != mod.module_dim:
            raise RuntimeError("inconsistent input shapes for AnswerHead.forward")
        dims = HeadDims(B, N, D, mod.c_pad, mod.dtype_code)
        L_.check(lib.dvgr_head_check_dims(C.byref(dims)), "dvgr_head_check_dims")
        dev = vis.device
        flat = mod.flat_params()
        logits = torch.empty((B, mod.c_pad), device=dev, dtype=torch.float32)
        saved = torch.empty(lib.dvgr_head_saved_bytes(C.byref(dims)), device=dev, dtype=torch.uint8)
        ws = torch.empty(lib.dvgr_head_workspace_bytes(C.byref(dims)), device=dev, dtype=torch.uint8)
        bn = mod.output_unit.classifier[3]
        train = bool(mod.training)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train else 0
        if mod.forced_seed is not None:
            seed = int(mod.forced_seed)
        io = HeadIO(vis.data_ptr(), qe.data_ptr(), logits.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        tflag = int(train) | (0 if mod.dropout_enabled else HEAD_TRAIN_NO_DROPOUT)
        L_.check(lib.dvgr_head_forward(C.byref(dims), flat.data_ptr(), C.byref(io), saved.data_ptr(), saved.numel(), ws.data_ptr(),
                                       ws.numel(), tflag, seed, stream), "dvgr_head_forward")
        if train:
            bn.num_batches_tracked += 1
        ctx.mod, ctx.dims, ctx.train, ctx.seed = mod, dims, tflag, seed
        ctx.save_for_backward(vis, qe, saved, flat)
        mod.last_logits_padded = logits
        return logits[:, :mod.num_answers]

    @staticmethod
    def backward(ctx, g_logits):
        lib = load()
        mod, dims = ctx.mod, ctx.dims
        vis, qe, saved, flat = ctx.saved_tensors
        dev = vis.device
        if (g_logits.dim() == 2 and g_logits.shape[1] == mod.num_answers and g_logits.stride(0) == mod.c_pad
                and g_logits.stride(1) == 1 and g_logits.storage_offset() == 0 and g_logits.dtype == torch.float32):
            gl = g_logits                       # already a view of a (B, c_pad) buffer (cross_entropy() below): no copy
        else:
            gl = torch.zeros((dims.B, mod.c_pad), device=dev, dtype=torch.float32)
            gl[:, :mod.num_answers] = g_logits
        g_vis, g_qe, gflat = torch.empty_like(vis), torch.empty_like(qe), torch.empty_like(flat)
        ws = torch.empty(lib.dvgr_head_workspace_bytes(C.byref(dims)), device=dev, dtype=torch.uint8)
        bn = mod.output_unit.classifier[3]
        io = HeadIO(vis.data_ptr(), qe.data_ptr(), None, bn.running_mean.data_ptr(), bn.running_var.data_ptr())
        gr = HeadGrads(gl.data_ptr(), g_vis.data_ptr(), g_qe.data_ptr(), gflat.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        L_.check(lib.dvgr_head_backward(C.byref(dims), flat.data_ptr(), C.byref(io), C.byref(gr), saved.data_ptr(), saved.numel(),
                                        ws.data_ptr(), ws.numel(), int(ctx.train), ctx.seed, stream), "dvgr_head_backward")
        if mod.grad_sink is not None:             # data parallel: see clip._LstmFn.backward
            mod.grad_sink(mod, gflat)
        pgrads = []
        for (o, n, shape, rows_used) in mod.param_slices:
            gview = gflat[o:o + n].view(shape)
            pgrads.append(gview if rows_used is None else gview[:rows_used])
        return (None, g_qe, g_vis, *pgrads)


class AnswerHead(nn.Module):
    """feature_aggregation (ContextSelfAttn) + output_unit (SimpleOutputUnitOpenEnded) of the reference's DualVGR."""

    def __init__(self, module_dim: int = 768, num_answers: int = 1000, dtype: str = "fp32"):
        super().__init__()
        self.module_dim, self.num_answers = int(module_dim), int(num_answers)
        self.c_pad = (self.num_answers + 3) // 4 * 4          # the last Linear is padded to a multiple of 4 classes
        self.dtype_code = {"fp32": L_.DVGR_F32, "bf16": L_.DVGR_BF16, "bf16x3": L_.DVGR_BF16X3}[dtype]
        D = self.module_dim
        self.feature_aggregation = _Readout(D)
        self.output_unit = _Classifier(D, self.num_answers)
        self.forced_seed = None
        self.dropout_enabled = True    # False: training-mode BatchNorm with the dropout sites off (DVGR_HEAD_TRAIN_NO_DROPOUT)
        self.last_logits_padded = None
        self.grad_sink = None          # callable(module, flat_gradient) or None (parallel.NativeParamReducer)
        cl = self.output_unit.classifier
        self._plist = [self.feature_aggregation.v_proj.weight, self.feature_aggregation.attn.weight,
                       self.feature_aggregation.attn.bias, self.output_unit.question_proj.weight,
                       self.output_unit.question_proj.bias, cl[1].weight, cl[1].bias, cl[3].weight, cl[3].bias, cl[5].weight,
                       cl[5].bias]
        self._kinds = [HP_VPROJ_W, HP_ATTN_W, HP_ATTN_B, HP_QPROJ_W, HP_QPROJ_B, HP_FC1_W, HP_FC1_B, HP_BN_W, HP_BN_B, HP_FC2_W,
                       HP_FC2_B]
        self._table = None
        self._flat = None
        self._ptrs = None
        with torch.no_grad():
            for p in self._plist:
                if p.dim() == 2:
                    nn.init.xavier_uniform_(p)          # model/utils.py:8-33 init_modules(w_init="xavier_uniform")
                elif p is not cl[3].weight:
                    p.zero_()
        self._build_table()

    def _build_table(self):
        # offsets do not depend on B / N: query them once with placeholder values
        dims = HeadDims(1, 1, self.module_dim, self.c_pad, self.dtype_code)
        try:
            lib = load()
        except RuntimeError:
            self._table = None           # library not built: the module can still hold / load parameters
            return
        tab = []
        for p, kind in zip(self._plist, self._kinds):
            off, rows, cols = C.c_int64(), C.c_int64(), C.c_int64()
            L_.check(lib.dvgr_head_param_offset(C.byref(dims), kind, C.byref(off), C.byref(rows), C.byref(cols)),
                     "dvgr_head_param_offset")
            tab.append(off.value)
        self._table = tab
        self._total = int(lib.dvgr_head_param_count(C.byref(dims)))

    @property
    def param_slices(self):
        out = []
        for p, off in zip(self._plist, self._table):
            out.append((off, p.numel(), tuple(p.shape), None))
        return out

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

    def cross_entropy(self, logits, target, scale: float = 1.0):
        """nn.CrossEntropyLoss()(logits, target) of train.py:147 on the logits this module just produced: returns
        (loss (1-element tensor), g_logits) from one fused launch; feed `g_logits` to
        `torch.autograd.backward([logits, ...], [g_logits, ...])`."""
        lib = load()
        lp = self.last_logits_padded
        B = lp.shape[0]
        dims = HeadDims(B, 1, self.module_dim, self.c_pad, self.dtype_code)
        loss = torch.empty(1, device=lp.device, dtype=torch.float32)
        g = torch.empty_like(lp)
        L_.check(lib.dvgr_head_cross_entropy(C.byref(dims), self.num_answers, lp.data_ptr(),
                                             target.to(torch.int64).contiguous().data_ptr(), float(scale), loss.data_ptr(),
                                             g.data_ptr(), torch.cuda.current_stream(lp.device).cuda_stream),
                 "dvgr_head_cross_entropy")
        return loss, g[:, :self.num_answers]

    def forward(self, question_embedding, visual_feature):
        """(B, D) question embedding, (B, N, D) visual_feature of the unit -> logits (B, num_answers)."""
        return _HeadFn.apply(self, question_embedding, visual_feature, *self._plist)
