"""Host side of the MI355X-native DualVGR reasoning unit.

`DualVGRUnit_multiple` keeps the constructor / forward() signature, the 7-tuple return and the
state_dict names of the reference module (model/models.py:86-173) but owns no arithmetic: the
forward and the backward of the whole stacked unit are the hand-written HIP kernels behind the C
ABI of include/dvgr.h (libdvgr_hip.so).  PyTorch is used for device memory, the current stream
and autograd bookkeeping only.  There is no eager fallback: without the HIP library, or with
tensors that are not on a ROCm device, the module raises.

Parameter storage: all parameters live in ONE flat fp32 buffer laid out for the kernels (the 8
head matrices of a modality are one contiguous (8F, D) block, etc. -- dvgr_param_offset); every
named nn.Parameter is a view into it, so optimizers update the kernel-side weights in place and
the gradient is one flat buffer (one RCCL all-reduce bucket per layer for data parallelism).

Deliberate deviations from the reference: the four list outputs stay on the
device (the reference moves them to the CPU, model/models.py:153-160); no hard-coded 'cuda:1';
graph_layers 1 .. 4 (every reference config uses 1; > 1 follows the reference's [i + j] module sharing, model/models.py:150-160).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Tuple

import torch
import torch.nn as nn

from . import _lib as L_

N_HEADS = 4
DROP_P = 0.15


def param_table(dims: L_.Dims) -> List[Tuple[str, Tuple[int, ...], int]]:
    """(reference state_dict name, shape, element offset in the flat buffer) for every parameter."""
    D, Wd, Fh, L = dims.D, dims.Wd, dims.D // 4, dims.L
    tab = []
    for l in range(L):
        o, _, _ = L_.param_offset(dims, L_.P_QA_W1, l)
        tab.append((f"queryAttn.{l}.feat_enhance.weight", (D, D), o))
        o, _, _ = L_.param_offset(dims, L_.P_QA_B1, l)
        tab.append((f"queryAttn.{l}.feat_enhance.bias", (D,), o))
        o, _, _ = L_.param_offset(dims, L_.P_QA_W2, l)
        tab.append((f"queryAttn.{l}.fc.weight", (1, D), o))
        o, _, _ = L_.param_offset(dims, L_.P_QA_B2, l)
        tab.append((f"queryAttn.{l}.fc.bias", (1,), o))
        for mod, (qp, com, spec, att) in enumerate((("queryPunish_appear", "acGCN", "appearance_GCN", "attention_appearance"),
                                                     ("queryPunish_motion", "mcGCN", "motion_GCN", "attention_motion"))):
            o, _, _ = L_.param_offset(dims, L_.P_WQ, l, mod)
            tab.append((f"{qp}.{l}.query_weight.weight", (D, Wd), o))
            o, _, _ = L_.param_offset(dims, L_.P_BQ, l, mod)
            tab.append((f"{qp}.{l}.query_weight.bias", (D,), o))
            # GAT modules: the reference builds L * G per family (model/models.py:95-100); module k < L sits in layer block
            # k, the others in the extra region behind the query block (csrc/dvgr_common.h)
            G = max(int(dims.G), 1)
            for k in ([l] if l < L - 1 else [l] + list(range(L, L * G))):
                ow, _, _ = L_.param_offset(dims, L_.P_GAT_W, k, mod)
                ob, _, _ = L_.param_offset(dims, L_.P_GAT_B, k, mod)
                oa, _, _ = L_.param_offset(dims, L_.P_GAT_A, k, mod)
                oc, _, _ = L_.param_offset(dims, L_.P_GAT_C, k, mod)
                for g, fam in enumerate((com, spec)):
                    for h in range(N_HEADS):
                        hh = g * N_HEADS + h
                        tab.append((f"{fam}.{k}.attention_{h}.W.weight", (Fh, D), ow + hh * Fh * D))
                        tab.append((f"{fam}.{k}.attention_{h}.W.bias", (Fh,), ob + hh * Fh))
                        tab.append((f"{fam}.{k}.attention_{h}.a.weight", (1, 2 * Fh), oa + hh * 2 * Fh))
                        tab.append((f"{fam}.{k}.attention_{h}.a.bias", (1,), oc + hh))
            o, _, _ = L_.param_offset(dims, L_.P_SF_P1, l, mod)
            tab.append((f"{att}.{l}.project.0.weight", (D, D), o))
            o, _, _ = L_.param_offset(dims, L_.P_SF_BP, l, mod)
            tab.append((f"{att}.{l}.project.0.bias", (D,), o))
            o, _, _ = L_.param_offset(dims, L_.P_SF_P2, l, mod)
            tab.append((f"{att}.{l}.project.2.weight", (1, D), o))
    for nm, kw, kb, shp in (("linear0", L_.P_MFB_W0, L_.P_MFB_B0, (512, D)), ("linear1", L_.P_MFB_W1, L_.P_MFB_B1, (512, D)),
                            ("linear_out", L_.P_MFB_WO, L_.P_MFB_BO, (D, 256))):
        o, _, _ = L_.param_offset(dims, kw)
        tab.append((f"visualfusion.{nm}.weight", shp, o))
        o, _, _ = L_.param_offset(dims, kb)
        tab.append((f"visualfusion.{nm}.bias", (shp[0],), o))
    return tab


class _Holder(nn.Module):
    """Parameter container with nn.Linear's attribute names (weight / bias); it never computes."""

    def __init__(self, out_f: int, in_f: int, bias: bool = True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_f))
        else:
            self.register_parameter("bias", None)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the arithmetic runs in the HIP kernels")


class _Head(nn.Module):
    def __init__(self, D, Fh):
        super().__init__()
        self.W = _Holder(Fh, D)
        self.a = _Holder(1, 2 * Fh)


class _GAT(nn.Module):
    def __init__(self, D, Fh):
        super().__init__()
        for h in range(N_HEADS):
            self.add_module(f"attention_{h}", _Head(D, Fh))


class _QueryAttn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.feat_enhance = _Holder(D, D)
        self.fc = _Holder(1, D)


class _QueryPunish(nn.Module):
    def __init__(self, Wd, D):
        super().__init__()
        self.query_weight = _Holder(D, Wd)


class _SFAttn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.project = nn.Sequential(_Holder(D, D), nn.Identity(), _Holder(1, D, bias=False))


class _MFB(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.linear0 = _Holder(512, D)
        self.linear1 = _Holder(512, D)
        self.linear_out = _Holder(D, 256)


class _UnitFn(torch.autograd.Function):
    """Forward / backward of the whole stacked unit through the C ABI."""

    @staticmethod
    def forward(ctx, mod: "DualVGRUnit_multiple", A, M, dq, words, qlen, anchor, *params):
        # `anchor` is a 1-element leaf that requires grad: it keeps this node in the autograd graph when
        # the parameter gradients are deposited directly (grad_mode='flat') and the inputs need no grad.
        lib = L_.load()
        if not A.is_cuda:
            raise RuntimeError("DualVGRUnit_multiple runs on a ROCm device only (no CPU fallback); move the module "
                               "and its inputs to 'cuda'")
        A, M, dq, words = (t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()
                           for t in (A, M, dq, words))
        if qlen.dtype != torch.int64 or qlen.device != A.device or not qlen.is_contiguous():
            qlen = qlen.to(device=A.device, dtype=torch.int64).contiguous()
        B, N, D = A.shape
        T, Wd = dq.shape[1], words.shape[2]
        if M.shape != A.shape or dq.shape != (B, T, D) or words.shape[:2] != (B, T) or qlen.shape != (B,):
            raise RuntimeError("inconsistent input shapes for DualVGRUnit_multiple.forward")
        if D != mod.module_dim or Wd != mod.word_dim:
            raise RuntimeError(f"feature sizes {D}/{Wd} do not match the module ({mod.module_dim}/{mod.word_dim})")
        dev = A.device
        st_ = mod._shape_state(B, N, T, dev)       # dims (checked once), buffer sizes, reusable buffers of this shape
        dims = st_["dims"]
        flat = mod.flat_params()
        Lr = mod.layers * mod.graph_layers          # list entries: one per (unit layer, graph layer)
        if mod.reuse_buffers:
            vis, aqe, mqe, lists, saved = mod._shape_buffers(st_, "fwd", lambda: (
                torch.empty((B, N, D), device=dev, dtype=torch.float32), torch.empty((B, N, D), device=dev, dtype=torch.float32),
                torch.empty((B, N, D), device=dev, dtype=torch.float32),
                [torch.empty((Lr, B, N, D), device=dev, dtype=torch.float32) for _ in range(4)],
                torch.empty(st_["saved_bytes"], device=dev, dtype=torch.uint8)))
        else:
            vis = torch.empty((B, N, D), device=dev, dtype=torch.float32)
            aqe = torch.empty_like(vis)
            mqe = torch.empty_like(vis)
            lists = [torch.empty((Lr, B, N, D), device=dev, dtype=torch.float32) for _ in range(4)]
            saved = torch.empty(st_["saved_bytes"], device=dev, dtype=torch.uint8)
        ws = mod.workspace(st_["ws_bytes"], dev)
        train = bool(mod.training) and mod.dropout_enabled     # the unit's only training-mode behaviour is dropout
        seed = mod._next_seed() if train else 0
        if mod.forced_seed is not None:
            seed = int(mod.forced_seed)
        io = L_.IO(A.data_ptr(), M.data_ptr(), dq.data_ptr(), words.data_ptr(), qlen.data_ptr(), vis.data_ptr(),
                   aqe.data_ptr(), mqe.data_ptr(), lists[0].data_ptr(), lists[1].data_ptr(), lists[2].data_ptr(),
                   lists[3].data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        opts = mod.make_opts(train, seed, backward=False)
        mod.last_loss = None
        if mod.loss_in_forward is not None:
            # scope-U loss and its gradients by the forward itself (dvgr_opts.loss): under the last contractions
            alpha, beta, scale = mod.loss_in_forward[:3]
            with_vis = len(mod.loss_in_forward) < 4 or bool(mod.loss_in_forward[3])   # False: auxiliary terms alone
            if mod.reuse_buffers:
                terms, g_vis, g_lists = mod._shape_buffers(st_, "loss" + str(int(with_vis)), lambda: (
                    torch.empty(3, device=dev, dtype=torch.float32), torch.empty_like(vis) if with_vis else None,
                    [torch.empty_like(t) for t in lists]))
            else:
                terms = torch.empty(3, device=dev, dtype=torch.float32)
                g_vis = torch.empty_like(vis) if with_vis else None
                g_lists = [torch.empty_like(t) for t in lists]
            spec = L_.LossSpec(float(alpha), float(beta), float(scale), terms.data_ptr(),
                               g_vis.data_ptr() if with_vis else None, *[g.data_ptr() for g in g_lists])
            opts.loss = C.pointer(spec)
            mod.last_loss = (terms, [g_vis, None, None, *g_lists])
        L_.check(lib.dvgr_unit_forward(C.byref(dims), flat.data_ptr(), mod.lowp_ptr(), C.byref(io), saved.data_ptr(),
                                       saved.numel(), ws.data_ptr(), ws.numel(), C.byref(opts), stream),
                 "dvgr_unit_forward")
        ctx.mod, ctx.dims, ctx.train, ctx.seed, ctx.shape_state = mod, dims, train, seed, st_
        ctx.set_materialize_grads(False)     # outputs nobody differentiates arrive as None = NULL in dvgr_grads
        ctx.save_for_backward(A, M, dq, words, qlen, vis, aqe, mqe, *lists, saved, flat)
        mod.last_saved = saved if mod.keep_saved else None
        return (vis, aqe, mqe, *lists)

    @staticmethod
    def backward(ctx, g_vis, g_aqe, g_mqe, g_ca, g_cm, g_aq, g_mq):
        lib = L_.load()
        mod, dims = ctx.mod, ctx.dims
        A, M, dq, words, qlen, vis, aqe, mqe, l0, l1, l2, l3, saved, flat = ctx.saved_tensors
        dev = A.device
        ups = [None if g is None else (g if (g.dtype == torch.float32 and g.is_contiguous()) else g.contiguous().float())
               for g in (g_vis, g_aqe, g_mqe, g_ca, g_cm, g_aq, g_mq)]
        st_ = ctx.shape_state
        # fresh tensors even with reuse_buffers: autograd's AccumulateGrad keeps an incoming gradient as .grad without a copy
        # only when nobody else holds a reference to it
        gA = torch.empty_like(A)
        gM = torch.empty_like(M)
        gdq = torch.empty_like(dq)
        gwords = torch.empty_like(words)
        flat_mode = mod.grad_mode == "flat"
        gflat, deposit = mod._grad_target(flat) if flat_mode else (torch.empty_like(flat), None)
        if deposit == "accumulate" and (mod._layer_cb is not None or getattr(mod, "_comm", None) is not None):
            # the callback hands `gflat` to an asynchronous reducer; adding a half-reduced temporary into the live
            # buffer afterwards would race with the collectives
            raise RuntimeError("gradient accumulation into a live flat buffer cannot be combined with the layer callback: "
                               "call zero_grad_flat() (or zero_grad(set_to_none=True)) before every backward")
        ws = mod.workspace(st_["ws_bytes"], dev)
        io = L_.IO(A.data_ptr(), M.data_ptr(), dq.data_ptr(), words.data_ptr(), qlen.data_ptr(), vis.data_ptr(),
                   aqe.data_ptr(), mqe.data_ptr(), l0.data_ptr(), l1.data_ptr(), l2.data_ptr(), l3.data_ptr())
        gr = L_.Grads(*[None if u is None else u.data_ptr() for u in ups],
                      gA.data_ptr(), gM.data_ptr(), gdq.data_ptr(), gwords.data_ptr(), gflat.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        opts = mod.make_opts(ctx.train, ctx.seed, backward=True, dims=dims)
        mod._backward_gflat = gflat
        L_.check(lib.dvgr_unit_backward(C.byref(dims), flat.data_ptr(), mod.lowp_ptr(), C.byref(io), C.byref(gr),
                                        saved.data_ptr(), saved.numel(), ws.data_ptr(), ws.numel(), C.byref(opts),
                                        stream), "dvgr_unit_backward")
        if getattr(mod, "_layer_cb_error", None) is not None:
            err, mod._layer_cb_error = mod._layer_cb_error, None
            raise err
        if flat_mode:
            mod._deposit_grad(gflat, deposit)
            return (None, gA, gM, gdq, gwords, None, None)
        mod.last_flat_grad = gflat
        pgrads = [gflat[o:o + n].view(shape) for (o, n, shape) in mod.param_slices]
        return (None, gA, gM, gdq, gwords, None, None, *pgrads)


_PLACED_DEVICES = set()      # devices whose internal streams were placed by this host side (dvgr_streams_place)


class DualVGRUnit_multiple(nn.Module):
    """Drop-in for the reference `DualVGRUnit_multiple` (model/models.py:86-173).

    Same constructor arguments and defaults (model/models.py:87).  Additions: `dtype` ('fp32' | 'bf16x3' |
    'bf16') selects the MFMA operand precision of the kernels (include/dvgr.h: DVGR_F32 exact fp32,
    DVGR_BF16X3 fp32-grade split-bf16, DVGR_BF16 plain bf16 operands); `grad_mode` selects how parameter
    gradients reach `.grad`:
      'flat'     (default) the backward writes ONE flat gradient buffer and every `p.grad` is a view
                 into it (no per-parameter kernels; `zero_grad_flat()` marks it reusable in place).
                 Works with `loss.backward()` + any optimizer; the parameters are not autograd inputs,
                 so `torch.autograd.grad(loss, params)` needs 'autograd'.
      'autograd' every parameter is an input of the autograd node (plain PyTorch semantics)."""

    def __init__(self, word_dim=300, module_dim=512, num_of_nodes=8, appearance_graph_layers=1,
                 motion_graph_layers=1, graph_module='GAT', unit_layers=3, dtype='fp32', grad_mode='flat'):
        super().__init__()
        if grad_mode not in ('flat', 'autograd'):
            raise ValueError("grad_mode must be 'flat' or 'autograd'")
        self.grad_mode = grad_mode
        if graph_module != 'GAT':
            # the reference builds no GNN lists for any other value and then fails in forward
            raise ValueError("graph_module must be 'GAT' (model/models.py:94)")
        if appearance_graph_layers != motion_graph_layers or not 1 <= appearance_graph_layers <= 4:
            # (DualVGR passes ONE graph_layers to both, model/models.py:44-45; the kernels batch the two modalities)
            raise NotImplementedError("appearance_graph_layers and motion_graph_layers must be equal and between 1 and 4")
        if module_dim % 64 or word_dim % 4:
            raise ValueError("module_dim must be a multiple of 64 and word_dim a multiple of 4")
        self.layers = unit_layers
        self.module_dim = module_dim
        self.word_dim = word_dim
        self.num_of_nodes = num_of_nodes
        self.num_of_appearance_graph_layers = appearance_graph_layers
        self.num_of_motion_graph_layers = motion_graph_layers
        self.dtype_code = {'fp32': L_.DVGR_F32, 'bf16': L_.DVGR_BF16, 'bf16x3': L_.DVGR_BF16X3}[dtype]
        D, Wd, Fh, Lr = module_dim, word_dim, module_dim // 4, unit_layers
        G = self.graph_layers = appearance_graph_layers
        self.queryAttn = nn.ModuleList([_QueryAttn(D) for _ in range(Lr)])
        self.queryPunish_appear = nn.ModuleList([_QueryPunish(Wd, D) for _ in range(Lr)])
        self.queryPunish_motion = nn.ModuleList([_QueryPunish(Wd, D) for _ in range(Lr)])
        self.appearance_GCN = nn.ModuleList([_GAT(D, Fh) for _ in range(Lr * G)])
        self.motion_GCN = nn.ModuleList([_GAT(D, Fh) for _ in range(Lr * G)])
        self.acGCN = nn.ModuleList([_GAT(D, Fh) for _ in range(Lr * G)])
        self.mcGCN = nn.ModuleList([_GAT(D, Fh) for _ in range(Lr * G)])
        self.attention_appearance = nn.ModuleList([_SFAttn(D) for _ in range(Lr)])
        self.attention_motion = nn.ModuleList([_SFAttn(D) for _ in range(Lr)])
        self.visualfusion = _MFB(D)
        # normalised all-ones adjacency, kept for API compatibility only (it is strictly positive, so
        # the reference's `adj > 0` mask never fires: model/models.py:114-119, model/GraphNN.py:100-101)
        adj = torch.full((num_of_nodes, num_of_nodes), 1.0 / (num_of_nodes + 1))
        adj += torch.eye(num_of_nodes) / (num_of_nodes + 1)
        self.appearance_adj = adj
        self.motion_adj = adj.clone()

        self._dims0 = L_.make_dims(1, num_of_nodes, 1, D, Wd, Lr, G, self.dtype_code)
        L_.load()       # raises with the build instructions when libdvgr_hip.so is missing
        self._table = param_table(self._dims0)
        named = dict(self.named_parameters())
        assert set(named) == {n for n, _, _ in self._table}, "parameter inventory mismatch"
        self._names = [n for n, _, _ in self._table]
        self._plist = [named[n] for n in self._names]      # Parameter objects are stable across .to()
        self.param_slices = [(o, int(torch.tensor(s).prod()), s) for _, s, o in self._table]
        self._flat = None
        self._lowp = None
        self._lowp_version = None
        self._p1t = None
        self._lo8 = None
        self._p1t_stale = True
        self._ws = None
        self._events = None
        self._events_arr = None
        self._probe = None
        self._layer_cb = None
        self._comm = None
        self._layer_cb_c = None
        self._layer_cb_error = None
        self._backward_gflat = None
        self.no_planes = False       # diagnostics: bf16 modes convert inside the contraction kernels instead
        self.forced_seed = None      # tests: fix the dropout seed of the next forward
        self.dropout_enabled = True  # False: training mode without the 24 dropout sites per layer
        self.keep_saved = False      # tests: keep a handle on the saved-for-backward buffer
        # True: outputs, the saved-for-backward buffer, the loss gradients and the input gradients of one (B, N, T) shape
        # are allocated ONCE and rewritten by every step (a training loop that consumes a step's tensors before the next
        # forward -- trainer.TrainStep, bench.py -- loses nothing and saves ~20 allocator calls + 0.36 GB of fresh memory
        # per step at the benchmark shape).  False (default): fresh tensors per call, plain PyTorch semantics.
        self.reuse_buffers = False
        self._shapes = {}
        self.single_stream = bool(int(os.environ.get("DVGR_SINGLE_STREAM", "0")))   # DVGR_OPT_SINGLE_STREAM
        self.fused_gat = not bool(int(os.environ.get("DVGR_NO_FUSED_GAT", "0")))      # False: DVGR_OPT_NO_FUSED_GAT (two launches)
        # False: DVGR_OPT_NO_FUSED_GAT_BWD (dz contraction, then attention backward); "force": wherever the shape is supported,
        # not only where it is measured faster (DVGR_OPT_FORCE_FUSED_GAT_BWD)
        self.fused_gat_bwd = not bool(int(os.environ.get("DVGR_NO_FUSED_GAT_BWD", "0")))
        if self.fused_gat_bwd and os.environ.get("DVGR_FORCE_FUSED_GAT_BWD", "0") == "1":
            self.fused_gat_bwd = "force"
        self.forks_by_record = bool(int(os.environ.get("DVGR_FORKS_BY_RECORD", "0")))       # True: DVGR_OPT_FORKS_BY_RECORD
        self.dp_on_main = bool(int(os.environ.get("DVGR_DP_ON_MAIN", "0")))                 # True: DVGR_OPT_DP_ON_MAIN (d p launches of the layers >= 1 on the caller's stream)
        self.s0_where = os.environ.get("DVGR_S0", "split")         # "split" (default) | "main" (DVGR_OPT_S0_ON_MAIN) | "side" (DVGR_OPT_S0_ON_SIDE: rounds 1-5)
        self.folded_punish = not bool(int(os.environ.get("DVGR_NO_FOLDED_PUNISH", "0")))     # False: DVGR_OPT_NO_FOLDED_PUNISH (the S1 backward's dX half as a launch of its own)
        self.half_gat_tiles = not bool(int(os.environ.get("DVGR_NO_HALF_GAT_TILES", "0")))   # False: DVGR_OPT_NO_HALF_GAT_TILES (64-row tiles of the fused kernels at every batch)
        self.kb64 = not bool(int(os.environ.get("DVGR_KB32", "0")))                    # False: DVGR_OPT_KB32 (32-deep K-tiles in the fused kernels)
        # 8-bit lo planes of the fused kernels' weight operands (dvgr_opts.lo8): an experiment of round 5 that applies to the
        # 32-deep K-tiles only and bought nothing there (fewer bytes, the same number of L2 requests); off unless asked for
        self.lo8_planes = bool(int(os.environ.get("DVGR_LO8", "0")))
        self.hoist_p1t = not bool(int(os.environ.get("DVGR_P1T_PER_CALL", "0")))     # False: the backward transposes P1's planes itself, per call
        self.s5_reads_tn = bool(int(os.environ.get("DVGR_S5_READS_TN", "0")))                # DVGR_OPT_S5_READS_TN (A/B, parity)
        self.fused_bwd_reads_dz = bool(int(os.environ.get("DVGR_FUSED_BWD_READS_DZ", "0")))   # DVGR_OPT_FUSED_BWD_READS_DZ (A/B, parity)
        self.leaves_on_query = bool(int(os.environ.get("DVGR_LEAVES_ON_QUERY", "0")))  # True: DVGR_OPT_LEAVES_ON_QUERY (A/B timing)
        # (alpha, beta, dep_scale[, with_sum_vis = True]) or None: when set, forward_raw() also evaluates the scope-U loss
        # sum(vis) + alpha*common + beta*dep_scale*dependence with its gradients w.r.t. the outputs (dvgr_opts.loss)
        # and leaves (terms, upstream_grads) in `last_loss`, ready for torch.autograd.backward(raw, upstream_grads)
        self.loss_in_forward = None
        self.last_loss = None
        self._seed_pool = []
        self.last_saved = None
        self.last_flat_grad = None
        self._grad_flat = None       # persistent flat gradient buffer (grad_mode='flat')
        self._grad_fresh = False     # True: the next backward may overwrite _grad_flat in place
        self._anchor = torch.zeros(1, requires_grad=True)
        self._reset_parameters()
        self._flatten()

    # ---- flat gradients ----------------------------------------------------------------------------------
    def _grads_alias_flat(self) -> bool:
        gf = self._grad_flat
        if gf is None:
            return False
        for idx in (0, -1):
            p, (o, _, _) = self._plist[idx], self.param_slices[idx]
            if p.grad is None or p.grad.data_ptr() != gf.data_ptr() + 4 * o:
                return False
        return True

    def _grad_target(self, flat):
        """Buffer the backward writes into and what to do afterwards: 'inplace' (persistent buffer,
        views already installed), 'accumulate' (add a temporary into it) or 'install' (new buffer)."""
        if self._grads_alias_flat() and self._grad_flat.device == flat.device:
            if self._grad_fresh:
                return self._grad_flat, "inplace"
            return torch.empty_like(flat), "accumulate"
        return torch.empty_like(flat), "install"

    def _deposit_grad(self, gflat, how):
        if how == "accumulate":
            self._grad_flat.add_(gflat)
        elif how == "install":
            self._grad_flat = gflat
            for p, (o, n, shape) in zip(self._plist, self.param_slices):
                p.grad = gflat[o:o + n].view(shape)
        self._grad_fresh = False
        self.last_flat_grad = self._grad_flat

    def zero_grad_flat(self):
        """Fast replacement of optimizer.zero_grad() for grad_mode='flat': no memset, no re-aliasing --
        the next backward simply overwrites the persistent flat gradient buffer (it writes every
        element).  Until then the old values stay readable."""
        self._grad_fresh = True

    # ---- initialisation as the reference constructs the same sub-modules --------------------------
    def _reset_parameters(self):
        named = dict(self.named_parameters())
        for name, p in named.items():
            with torch.no_grad():
                if name.startswith("visualfusion"):
                    # model/init_weight.py:4-13 via model/fusions/fusions.py:412-417
                    if name.endswith("weight"):
                        nn.init.normal_(p, mean=0.0, std=0.01)
                    else:
                        nn.init.zeros_(p)
                elif ".attention_" in name and name.endswith("weight"):
                    nn.init.xavier_uniform_(p, gain=1.414)      # model/GraphNN.py:90,92
                elif name.endswith("weight"):
                    nn.init.kaiming_uniform_(p, a=5 ** 0.5)     # nn.Linear default
                else:
                    fan_in = named[name[:-4] + "weight"].shape[1]
                    bound = 1.0 / fan_in ** 0.5
                    nn.init.uniform_(p, -bound, bound)

    # ---- flat storage ----------------------------------------------------------------------------------
    def _flatten(self):
        dev = self._plist[0].device
        total = L_.load().dvgr_param_count(C.byref(self._dims0))
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, (name, shape, off) in zip(self._plist, self._table):
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1).float())
            p.data = flat[off:off + n].view(shape)
        self._flat = flat
        self._ptrs = tuple(p.data_ptr() for p in self._plist)
        self._lowp = None
        self._lowp_version = None
        self._p1t = None
        self._lo8 = None
        self._p1t_stale = True

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if getattr(self, "_table", None) is not None:
            self._flatten()     # .to()/.cuda() gave every parameter its own storage: re-alias them
        return out

    def _shape_state(self, B, N, T, dev):
        key = (B, N, T, dev.index)
        st_ = self._shapes.get(key)
        if st_ is None:
            lib = L_.load()
            dims = L_.make_dims(B, N, T, self.module_dim, self.word_dim, self.layers, self.graph_layers, self.dtype_code)
            L_.check(lib.dvgr_check_dims(C.byref(dims)), "dvgr_check_dims")
            if dev.type == "cuda" and not self.single_stream and dev.index not in _PLACED_DEVICES:
                # set-up, not the hot path: the library's side / query streams get their hardware queues HERE (a device
                # synchronise + ~0.1 s of probe kernels, include/dvgr.h dvgr_streams_place) instead of inside the first
                # forward call; skipped by the library itself while a stream capture is active
                with torch.cuda.device(dev):
                    L_.check(lib.dvgr_streams_place(torch.cuda.current_stream(dev).cuda_stream), "dvgr_streams_place")
                _PLACED_DEVICES.add(dev.index)
            st_ = {"dims": dims, "saved_bytes": lib.dvgr_saved_bytes(C.byref(dims)), "ws_bytes": lib.dvgr_workspace_bytes(C.byref(dims)),
                   "buf": {}}
            if len(self._shapes) > 16:
                self._shapes.clear()
            self._shapes[key] = st_
        return st_

    @staticmethod
    def _shape_buffers(st_, what, make):
        b = st_["buf"].get(what)
        if b is None:
            b = st_["buf"][what] = make()
        return b

    def flat_params(self) -> torch.Tensor:
        # someone may have re-pointed .data (an external load that replaced tensors): every parameter must still be a view
        # into the flat buffer (~70 L pointer compares: negligible against a step)
        for p, ptr in zip(self._plist, self._ptrs):
            if p.data_ptr() != ptr:
                self._flatten()
                break
        return self._flat

    def lowp_ptr(self):
        """bf16 hi / lo planes of the parameters (2 x param_count bf16) for the bf16 MFMA modes, refreshed
        whenever a parameter changed (optimizer step, load_state_dict).  None in fp32 mode."""
        if self.dtype_code == L_.DVGR_F32 or self.no_planes:
            return None
        flat = self._flat
        # an in-place update (optimizer step over any SUBSET of the parameters, load_state_dict, a no_grad edit) bumps the
        # version counter of every tensor it touches: the sum over ALL parameters detects each of them (~70 L integers per
        # forward); a caller that rewrites the buffer behind autograd's back (a raw kernel) calls params_changed()
        ver = sum(p._version for p in self._plist)
        if self._lowp is None or self._lowp.device != flat.device or ver != self._lowp_version:
            if self._lowp is None or self._lowp.device != flat.device:
                self._lowp = torch.empty(2 * flat.numel(), device=flat.device, dtype=torch.bfloat16)
            L_.check(L_.load().dvgr_pack_weights(C.byref(self._dims0), flat.data_ptr(), self._lowp.data_ptr(),
                                                 torch.cuda.current_stream(flat.device).cuda_stream), "dvgr_pack_weights")
            self._lowp_version = ver
            self._p1t_stale = True
        return self._lowp.data_ptr()

    def p1t_ptr(self):
        """Transposed planes of every layer's project[0] weight for the fused GAT-view backward (dvgr_opts.p1t,
        dvgr_pack_p1t), refreshed whenever the parameter planes were: by a repack here, or by a caller that was handed
        lowp_planes() to rewrite them (trainer.ClipAdam does so once per optimizer step).  With static weights -- a benchmark's
        timed loop, evaluation -- they are prepared once; the library transposes per call when this is None (fp32 mode)."""
        lp = self.lowp_ptr()
        if lp is None or self.graph_layers != 1:
            return None
        dev = self._flat.device
        if self._p1t is None or self._p1t.device != dev:
            n = L_.load().dvgr_p1t_elems(C.byref(self._dims0))
            if n == 0:
                return None
            self._p1t = torch.empty(n, device=dev, dtype=torch.bfloat16)
            self._p1t_stale = True
        if self._p1t_stale:
            st = torch.cuda.current_stream(dev).cuda_stream
            L_.check(L_.load().dvgr_pack_p1t(C.byref(self._dims0), lp, self._p1t.data_ptr(), st), "dvgr_pack_p1t")
            # ... and, from both, the 8-bit block-scaled lo planes of the fused kernels' weight operands (dvgr_opts.lo8)
            nb = L_.load().dvgr_lo8_bytes(C.byref(self._dims0))
            if nb and self.lo8_planes:
                if self._lo8 is None or self._lo8.device != dev:
                    self._lo8 = torch.empty(nb, device=dev, dtype=torch.uint8)
                L_.check(L_.load().dvgr_pack_lo8(C.byref(self._dims0), lp, self._p1t.data_ptr(), self._lo8.data_ptr(), st),
                         "dvgr_pack_lo8")
            else:
                self._lo8 = None
            self._p1t_stale = False
        return self._p1t.data_ptr()

    def lo8_ptr(self):
        """8-bit block-scaled lo planes of Wcat / P1^T for the fused GAT-view kernels (dvgr_opts.lo8, dvgr_pack_lo8; bf16x3 only),
        kept current together with p1t_ptr().  None: those kernels stream the bf16 lo planes."""
        if not self.lo8_planes or self.p1t_ptr() is None or self._lo8 is None:
            return None
        return self._lo8.data_ptr()

    def lowp_planes(self):
        """The bf16 hi / lo planes tensor of the flat parameter buffer (None in fp32 mode), allocated if need be; for a
        caller that rewrites it together with the parameters in one pass (trainer.ClipAdam: no repack follows such a step
        because the fused kernel bumps no version counter; anyone else calls params_changed()).

        CONTRACT: everything derived from the planes -- the transposed P1 planes the fused backward contracts against, the
        8-bit lo planes -- is rebuilt at the next backward only if this object KNOWS the planes were rewritten.  Each call
        of lowp_planes() counts as one rewrite (trainer.TrainStep calls it every step).  A caller that keeps the tensor and
        rewrites it again later must call planes_rewritten() after every such rewrite; otherwise the next backward multiplies
        dz by LAST step's P1^T without any error."""
        if self.dtype_code == L_.DVGR_F32 or self.no_planes:
            return None
        self.lowp_ptr()
        self._p1t_stale = True              # the caller rewrites the planes: their transposed copies follow at the next backward
        return self._lowp

    def planes_rewritten(self):
        """The planes tensor handed out by lowp_planes() was rewritten again (see its contract): the derived copies (P1^T,
        8-bit lo planes) are rebuilt at the next backward."""
        self._p1t_stale = True

    def params_changed(self):
        """The flat parameter buffer was modified behind autograd's back (no `_version` bump, e.g. by a raw kernel): the
        bf16 planes are repacked at the next forward.  trainer.ClipAdam needs no such call when it is handed
        lowp_planes(): it rewrites the planes itself, in the pass that updates the parameters."""
        self._lowp_version = None

    def _next_seed(self) -> int:
        """Dropout seed of one forward, drawn from torch's CPU generator (so torch.manual_seed reproduces a run) in
        batches of 256: one small tensor op per 256 steps instead of one per step on the launch path."""
        if not self._seed_pool:
            self._seed_pool = torch.randint(0, 2 ** 62, (256,)).tolist()
        return self._seed_pool.pop()

    def workspace(self, nbytes: int, dev) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        return self._ws

    def set_layer_events(self, events):
        """events: list of L+2 torch.cuda.Event, one per block of the flat gradient buffer (layer_param_range), recorded
        by the backward as the blocks become final: index L = the MFB block, L-1 .. 0 = the layer blocks, L+1 = the query
        block (last).  None disables."""
        if events is not None and len(events) != self.layers + 2:
            raise ValueError(f"expected {self.layers + 2} events (L layer blocks, the MFB block, the query block)")
        for ev in events or []:
            if not ev.cuda_event:
                ev.record()          # torch creates the hipEvent lazily at the first record
        self._events = events

    def set_layer_callback(self, fn):
        """fn(layer_index, flat_grad) is called on the host, from inside the backward call, right after the event of
        `layer_index` (set_layer_events; index L = the MFB block, L+1 = the query block) has been enqueued: the place to launch that slice's
        all-reduce (parallel.LayerBucketReducer.launch).  None disables."""
        self._layer_cb = fn
        self._layer_cb_error = None
        if fn is None:
            self._layer_cb_c = None
            return

        def _thunk(index, _user):
            try:
                fn(int(index), self._backward_gflat)
            except BaseException as e:          # cannot propagate through the C frame: re-raised after the call
                self._layer_cb_error = e

        self._layer_cb_c = L_.LAYER_DONE_CB(_thunk)

    def set_comm(self, comm):
        """comm: a `parallel.NativeUnitReducer` (or None).  The backward then enqueues the RCCL all-reduce of every block of
        the flat gradient buffer itself (dvgr_opts.comm, include/dvgr_comm.h): no events, no callback, no helper thread;
        the reducer's finish() makes the caller's stream wait for the averaged gradients."""
        self._comm = comm

    def set_probe(self, site: int, layer: int = 0, begin=None, end=None):
        """Bracket one launch site (enum dvgr_probe_site) of the following calls with two
        torch.cuda.Event(enable_timing=True); site 0 disables.  Used by bench.py."""
        for ev in (begin, end):
            if site and not ev.cuda_event:
                ev.record()          # torch creates the hipEvent lazily at the first record
        self._probe = (int(site), int(layer), begin, end) if site else None

    def make_opts(self, train: bool, seed: int, backward: bool, dims=None) -> "L_.Opts":
        o = L_.Opts()
        o.train, o.seed = int(bool(train)), int(seed)
        if self.single_stream:
            o.flags |= L_.OPT_SINGLE_STREAM
        if not self.fused_gat:
            o.flags |= L_.OPT_NO_FUSED_GAT
        if not self.fused_gat_bwd:
            o.flags |= L_.OPT_NO_FUSED_GAT_BWD
        elif self.fused_gat_bwd == "force":
            o.flags |= L_.OPT_FORCE_FUSED_GAT_BWD
        if self.fused_bwd_reads_dz:
            o.flags |= L_.OPT_FUSED_BWD_READS_DZ
        if self.s5_reads_tn:
            o.flags |= L_.OPT_S5_READS_TN
        if self.leaves_on_query:
            o.flags |= L_.OPT_LEAVES_ON_QUERY
        if self._events or getattr(self, "_layer_cb", None) is not None:
            o.flags |= L_.OPT_NO_QUERY_STREAM      # a reducer's streams run beside the call (include/dvgr.h)
        if self._probe is not None:
            site, layer, b, e = self._probe
            if (site >= L_.PROBE_BWD_SF_DW) == backward:
                o.probe_site, o.probe_layer = site, layer
                o.probe_begin, o.probe_end = b.cuda_event, e.cuda_event
        if backward and getattr(self, "_comm", None) is not None:
            o.comm = self._comm.handle
        if not self.kb64:
            o.flags |= L_.OPT_KB32
        if not self.half_gat_tiles:
            o.flags |= L_.OPT_NO_HALF_GAT_TILES
        if not self.folded_punish:
            o.flags |= L_.OPT_NO_FOLDED_PUNISH
        if self.forks_by_record:
            o.flags |= L_.OPT_FORKS_BY_RECORD
        if self.dp_on_main:
            o.flags |= L_.OPT_DP_ON_MAIN
        if self.s0_where == "main":
            o.flags |= L_.OPT_S0_ON_MAIN
        elif self.s0_where == "side":
            o.flags |= L_.OPT_S0_ON_SIDE
        if not self.lo8_planes:
            o.flags |= L_.OPT_NO_LO8
        if backward and self.fused_gat_bwd and self.hoist_p1t and dims is not None and \
                L_.load().dvgr_backward_uses_p1t(C.byref(dims), o.flags):
            o.p1t = self.p1t_ptr()
        if self.lo8_planes and self.hoist_p1t and self.fused_gat and (not backward or o.p1t):
            o.lo8 = self.lo8_ptr()
        if backward and self._events:
            arr = (C.c_void_p * (self.layers + 2))()
            for i, ev in enumerate(self._events):
                arr[i] = ev.cuda_event
            self._events_arr = arr               # keep alive for the duration of the call
            o.layer_done_events = C.cast(arr, C.POINTER(C.c_void_p))
            if getattr(self, "_layer_cb_c", None) is not None:
                o.layer_done_cb = C.cast(self._layer_cb_c, C.c_void_p)
        return o

    def layer_param_range(self, layer: int) -> Tuple[int, int]:
        a, b = C.c_int64(), C.c_int64()
        L_.check(L_.load().dvgr_param_layer_range(C.byref(self._dims0), layer, C.byref(a), C.byref(b)),
                 "dvgr_param_layer_range")
        return a.value, b.value

    # ---- forward: the reference signature (model/models.py:121) ---------------------------------------
    def forward(self, appearance_video_feat, motion_video_feat, dynamic_question_embedding, word_embedding,
                question_len):
        vis, aqe, mqe, ca, cm, aq, mq = self.forward_raw(appearance_video_feat, motion_video_feat,
                                                         dynamic_question_embedding, word_embedding, question_len)
        return (vis, aqe, mqe, list(ca.unbind(0)), list(cm.unbind(0)), list(aq.unbind(0)), list(mq.unbind(0)))

    def forward_raw(self, appearance_video_feat, motion_video_feat, dynamic_question_embedding, word_embedding,
                    question_len):
        """Same as forward() but the four lists come back as (L, B, N, D) tensors (what the kernels write)."""
        params = self._plist if self.grad_mode == "autograd" else ()
        return _UnitFn.apply(self, appearance_video_feat, motion_video_feat, dynamic_question_embedding,
                             word_embedding, question_len, self._anchor, *params)
