"""The training step around the path: host-side mirror of one iteration of the reference's
loop, train.py:139-159 --

    optimizer.zero_grad()
    logits, aq_embed, mq_embed, com_app, com_motion, aq_fusion, mq_fusion = model(*batch_input)
    loss = CE(logits, answers) + alpha * mean_l common_loss(com_app[l], com_motion[l])
                               + beta * mean_l [loss_dependence(aq_fusion[l], com_app[l]) + loss_dependence(mq_fusion[l], com_motion[l])]
    loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), max_norm=12)
    optimizer.step()                                  # optim.Adam(model.parameters(), cfg.train.lr), train.py:85

-- with every piece on this package's HIP kernels: the auxiliary losses and their gradients inside the unit's forward
(dvgr_opts.loss), the cross-entropy fused with its gradient (dvgr_head_cross_entropy), and clip + Adam as two launches over
flat segments (include/dvgr_optim.h) instead of PyTorch's per-tensor loops.  `step_decay()` is train.py:341-349.

Data parallel (one process per GPU): pass `world` and the reducers; every rank's losses are means over its shard, the
batch-SUM HSIC term is multiplied by `world`, and gradients are AVERAGED before the clip, so the
step equals the single-process step on the concatenated batch.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib as L_


class OptimSeg(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("planes", C.c_void_p), ("lo_off", C.c_int64)]


OPTIM_SYMBOLS = [
    ("dvgr_optim_grad_sqnorm", C.c_int, [C.POINTER(OptimSeg), C.c_int, C.c_void_p, C.c_void_p]),
    ("dvgr_optim_adam_step", C.c_int, [C.POINTER(OptimSeg), C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_int64, C.c_void_p]),
]
_bound = False


def load():
    global _bound
    lib = L_.load()
    if not _bound:
        for name, restype, argtypes in OPTIM_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = True
    return lib


class ClipAdam:
    """`clip_grad_norm_(params, max_norm)` + `torch.optim.Adam(params, lr).step()` (train.py:85, :158-159) over flat
    segments.  A segment is (parameter tensor, gradient tensor) of equal numel; the unit contributes ONE (its flat buffers)."""

    def __init__(self, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 12.0):
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_norm)
        self.step_count = 0
        self._state: Dict[object, tuple] = {}       # segment key -> (exp_avg, exp_avg_sq), flat fp32
        self._layouts: Dict[object, str] = {}       # segment key -> layout tag of a flat multi-parameter segment
        self._sqnorm: Optional[torch.Tensor] = None
        self._arr = None

    def set_layout(self, key, tag: str) -> None:
        """Name the element layout of a FLAT segment (the unit's buffer: a digest of its parameter table).  The tag travels
        with state_dict(); load_state_dict() of moments recorded under another layout is refused instead of applying them
        to the wrong parameters (an equal numel proves nothing)."""
        self._layouts[key] = str(tag)

    def _moments(self, key, like: torch.Tensor):
        st = self._state.get(key)
        if st is None:
            st = (torch.zeros(like.numel(), device=like.device, dtype=torch.float32),
                  torch.zeros(like.numel(), device=like.device, dtype=torch.float32))
            self._state[key] = st
        elif st[0].numel() != like.numel():
            raise RuntimeError(f"ClipAdam: segment {key!r} has {like.numel()} elements but its stored moments have "
                               f"{st[0].numel()} (restored from a checkpoint of another model / layout?)")
        elif st[0].device != like.device:            # restored with map_location='cpu', or the model moved: keep the moments
            st = (st[0].to(like.device), st[1].to(like.device))
            self._state[key] = st
        return st

    def step(self, segments: List[tuple]) -> torch.Tensor:
        """segments: [(key, param, grad[, planes])], fp32, contiguous, on one ROCm device; `planes` (optional) is the bf16
        hi / lo planes tensor of that parameter buffer (2 x numel bf16), rewritten from the updated values in the same
        pass.  Returns the 1-element device tensor holding the SQUARE of the total gradient norm before clipping (no
        host sync)."""
        lib = load()
        if not segments:
            raise ValueError("no parameters to update")
        dev = segments[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("ClipAdam runs on a ROCm device only (no CPU fallback)")
        n = len(segments)
        if self._arr is None or len(self._arr) != n:
            self._arr = (OptimSeg * n)()
        for i, seg in enumerate(segments):
            key, p, g = seg[:3]
            planes = seg[3] if len(seg) > 3 else None
            if g is None:
                raise RuntimeError(f"segment {key!r} has no gradient")
            if p.numel() != g.numel() or not p.is_contiguous() or not g.is_contiguous() or p.dtype != torch.float32 \
                    or g.dtype != torch.float32:
                raise RuntimeError(f"segment {key!r}: parameter and gradient must be contiguous fp32 of equal size")
            m, v = self._moments(key, p)
            s = self._arr[i]
            s.param, s.grad, s.exp_avg, s.exp_avg_sq, s.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
            s.planes, s.lo_off = (planes.data_ptr(), p.numel()) if planes is not None else (None, 0)
        if self._sqnorm is None or self._sqnorm.device != dev:
            self._sqnorm = torch.zeros(1, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        self.step_count += 1
        L_.check(lib.dvgr_optim_grad_sqnorm(self._arr, n, self._sqnorm.data_ptr(), stream), "dvgr_optim_grad_sqnorm")
        L_.check(lib.dvgr_optim_adam_step(self._arr, n, self._sqnorm.data_ptr(), self.max_norm, self.lr, self.betas[0],
                                          self.betas[1], self.eps, self.step_count, stream), "dvgr_optim_adam_step")
        return self._sqnorm

    def state_dict(self):
        return {"format": "dvgr-clipadam-2", "step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "max_norm": self.max_norm, "layouts": dict(self._layouts),
                "state": {k: (m.clone(), v.clone()) for k, (m, v) in self._state.items()}}

    def load_state_dict(self, sd):
        for k, tag in sd.get("layouts", {}).items():
            if k in self._layouts and self._layouts[k] != tag:
                raise RuntimeError(f"ClipAdam: the moments of segment {k!r} were recorded under layout {tag}, this build lays the "
                                   f"segment out as {self._layouts[k]}; restore through TrainStep.load_state_dict (per-parameter)")
        if sd.get("format") != "dvgr-clipadam-2" and any(k in self._layouts for k in sd["state"]):
            raise RuntimeError("ClipAdam: checkpoint without layout tags holds a flat multi-parameter segment; its element order is "
                               "unknown -- restore through TrainStep.load_state_dict (torch.optim.Adam format) instead")
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        self.betas, self.eps, self.max_norm = tuple(sd["betas"]), float(sd["eps"]), float(sd["max_norm"])
        self._state = {k: (m.clone(), v.clone()) for k, (m, v) in sd["state"].items()}     # moved to the device at first use


class TrainStep:
    """One call = one iteration of train.py:139-159 on a `DualVGR` (this package's).  Returns the loss terms as device
    tensors (nothing is synchronised): {'loss', 'ce', 'common', 'dependence', 'grad_sqnorm', 'logits'}."""

    def __init__(self, model, lr: float = 1e-4, alpha: float = 1.0, beta: float = 1e-8, max_norm: float = 12.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, world: int = 1, unit_reducer=None, param_reducer=None):
        self.model = model
        self.unit = model.visual_input_unit
        self.head = model._head
        self.alpha, self.beta, self.world = float(alpha), float(beta), int(world)
        self.opt = ClipAdam(lr=lr, betas=betas, eps=eps, max_norm=max_norm)
        self.unit_reducer, self.param_reducer = unit_reducer, param_reducer
        self._unit_shared = getattr(self.unit, "graph_layers", 1) > 1
        if self._unit_shared and unit_reducer is not None:
            if not hasattr(unit_reducer, "reduce"):
                raise ValueError("graph_layers > 1 needs a reducer with a whole-buffer reduce(flat) (LayerBucketReducer / NativeUnitReducer)")
            # per-block events / callbacks / an in-call communicator cannot be honoured (DVGR_E_DIMS): the step reduces afterwards
            self.unit.set_layer_events(None)
            self.unit.set_layer_callback(None)
            self.unit.set_comm(None)
        if self.unit.grad_mode != "flat":
            raise ValueError("TrainStep needs the unit's flat gradient buffer (grad_mode='flat')")
        unit_ids = {id(p) for p in self.unit.parameters()}
        self.other = [(n, p) for n, p in model.named_parameters() if id(p) not in unit_ids and p.requires_grad]
        self.timing = None                           # bench.py: {'begin': Event, 'end': Event} around clip + Adam
        import hashlib
        self.opt.set_layout("visual_input_unit", hashlib.sha256(repr([(n, tuple(sh), int(o)) for n, sh, o in self.unit._table])
                                                                .encode()).hexdigest()[:16])

    @property
    def lr(self):
        return self.opt.lr

    def step_decay(self):
        """train.py:341-349: halve the learning rate."""
        self.opt.lr *= 0.5
        return self.opt.lr

    # ---- checkpointing: train.py:360-365 saves `optimizer.state_dict()`, train.py:119 restores it -------------------------
    def _unit_prefix(self) -> str:
        for name, mod in self.model.named_modules():
            if mod is self.unit:
                return name + "."
        raise RuntimeError("the unit is not a submodule of the model")

    def state_dict(self):
        """The optimiser state in `torch.optim.Adam.state_dict()` FORMAT over `model.parameters()` order ('state': index ->
        {'step', 'exp_avg', 'exp_avg_sq'} shaped like the parameter; one param group): what train.py:364 stores under
        'optimizer', independent of this package's flat layout.  Parameters that have not been stepped yet are absent,
        as in torch."""
        names = [n for n, _ in self.model.named_parameters()]
        shapes = {n: tuple(p.shape) for n, p in self.model.named_parameters()}
        pre = self._unit_prefix()
        off = {pre + n: (int(o), tuple(sh)) for n, sh, o in self.unit._table}
        state = {}
        flat = self.opt._state.get("visual_input_unit")
        for i, n in enumerate(names):
            if n in off:
                if flat is None:
                    continue
                o, sh = off[n]
                k = 1
                for d in sh:
                    k *= d
                mom = (flat[0][o:o + k].view(sh).clone(), flat[1][o:o + k].view(sh).clone())
            else:
                st = self.opt._state.get(n)
                if st is None:
                    continue
                mom = (st[0].view(shapes[n]).clone(), st[1].view(shapes[n]).clone())
            state[i] = {"step": torch.tensor(float(self.opt.step_count)), "exp_avg": mom[0], "exp_avg_sq": mom[1]}
        group = {"lr": self.opt.lr, "betas": self.opt.betas, "eps": self.opt.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "max_norm": self.opt.max_norm}

    def load_state_dict(self, sd):
        """Restore from `state_dict()` or from a reference checkpoint's `optimizer` entry (torch.optim.Adam over the
        reference model's parameters: same names, same order).  Moments may live on any device (map_location='cpu')."""
        names = [n for n, _ in self.model.named_parameters()]
        params = dict(self.model.named_parameters())
        group = sd["param_groups"][0]
        if len(group["params"]) != len(names):
            raise RuntimeError(f"optimizer state covers {len(group['params'])} parameters, the model has {len(names)}")
        pre = self._unit_prefix()
        off = {pre + n: (int(o), tuple(sh)) for n, sh, o in self.unit._table}
        flat_n = self.unit.flat_params().numel()
        dev = self.unit.flat_params().device
        flat = (torch.zeros(flat_n, device=dev), torch.zeros(flat_n, device=dev))
        new_state, steps, any_unit = {}, set(), False
        for pos, idx in enumerate(group["params"]):
            st = sd["state"].get(idx)
            if st is None:
                continue
            n = names[pos]
            if tuple(st["exp_avg"].shape) != tuple(params[n].shape):
                raise RuntimeError(f"optimizer state of {n}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(params[n].shape)}")
            steps.add(int(float(st["step"])))
            if n in off:
                o, sh = off[n]
                k = st["exp_avg"].numel()
                flat[0][o:o + k] = st["exp_avg"].reshape(-1).to(dev, torch.float32)
                flat[1][o:o + k] = st["exp_avg_sq"].reshape(-1).to(dev, torch.float32)
                any_unit = True
            else:
                new_state[n] = (st["exp_avg"].reshape(-1).to(params[n].device, torch.float32).clone(),
                                st["exp_avg_sq"].reshape(-1).to(params[n].device, torch.float32).clone())
        if len(steps) > 1:
            raise RuntimeError(f"per-parameter step counts differ ({sorted(steps)}): ClipAdam keeps ONE step count")
        if any_unit:
            new_state["visual_input_unit"] = flat
        self.opt._state = new_state
        self.opt.step_count = steps.pop() if steps else 0
        self.opt.lr, self.opt.betas, self.opt.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        if "max_norm" in sd:
            self.opt.max_norm = float(sd["max_norm"])

    def __call__(self, video_appearance_feat, video_motion_feat, question, question_len, answers):
        m, unit, head = self.model, self.unit, self.head
        # optimizer.zero_grad() (train.py:139): the unit's flat buffer is simply overwritten by the next backward
        unit.zero_grad_flat()
        for _, p in self.other:
            p.grad = None
        # the batch-SUM HSIC term is scaled by world under gradient averaging; the CE / common terms are means
        unit.loss_in_forward = (self.alpha, self.beta, float(self.world), False)
        if self.unit_reducer is not None:
            self.unit_reducer.start(unit.layers)
        if self.param_reducer is not None:
            self.param_reducer.start()
        logits, *_ = m(video_appearance_feat, video_motion_feat, question, question_len)
        ce, g_logits = head.cross_entropy(logits, answers)                   # train.py:147, value + gradient in one launch
        terms, g = unit.last_loss                                            # {0, mean_l common, mean_l dependence}, upstream grads
        raw = m.last_raw                                                     # (L, B, N, D) list tensors as the kernels wrote them
        torch.autograd.backward([logits, raw[3], raw[4], raw[5], raw[6]], [g_logits, g[3], g[4], g[5], g[6]])
        if self.unit_reducer is not None and not self._unit_shared:
            self.unit_reducer.finish()
        if self.param_reducer is not None:
            self.param_reducer.finish()
        if self.unit_reducer is not None and self._unit_shared:
            # graph_layers > 1: GAT modules are shared between unit layers, no block of the flat buffer is final before the
            # backward ends (the library refuses layer events / an in-call communicator there): ONE whole-buffer all-reduce
            # behind the backward, before clip + Adam -- issued from this thread AFTER the parameter buckets have all been
            # enqueued by the issuer thread (finish() above joins it), so every rank issues the same sequence
            self.unit_reducer.reduce(unit.last_flat_grad)
        # clip_grad_norm_(model.parameters(), 12) + Adam (train.py:158-159)
        segs = [("visual_input_unit", unit.flat_params(), unit.last_flat_grad, unit.lowp_planes())]
        for name, p in self.other:
            if p.grad is not None:
                segs.append((name, p.data, p.grad))
        if self.timing is not None:
            self.timing["begin"].record()
        sq = self.opt.step(segs)
        if self.timing is not None:
            self.timing["end"].record()
        loss = ce[0] + self.alpha * terms[1] + self.beta * terms[2]
        return {"loss": loss, "ce": ce[0], "common": terms[1], "dependence": terms[2], "grad_sqnorm": sq[0], "logits": logits}
