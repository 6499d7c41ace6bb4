"""Data parallelism for the unit: one process per GPU, gradients all-reduced over RCCL (xGMI).

The path shards by independent QA pairs: every rank runs the same kernels on
its own slice of the batch and the only exchange step is the gradient all-reduce.  Because the
backward writes ONE flat gradient buffer whose layer blocks become final in the order
MFB, L-1, ..., 0, the natural bucket is a layer block: `LayerBucketReducer` starts the all-reduce of
a block on a side stream as soon as the backward has recorded that block's event
(dvgr_opts.layer_done_events), overlapping communication with the remaining backward kernels.

Backends: "nccl" (= RCCL on ROCm) on GPUs; the same code runs over "gloo" on CPU tensors, which is
how tests/test_parallel_gloo.py covers it without GPUs.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def block_order(num_layers: int) -> List[int]:
    """Block indices (dvgr_param_layer_range) in the order the backward finalises them: the MFB block (L), the layer
    blocks L-1 .. 0, then the query block (L+1: QueryAttn / QueryPunish parameters of all layers)."""
    L = num_layers
    return [L] + list(range(L - 1, -1, -1)) + [L + 1]


def layer_buckets(module) -> List[Tuple[int, int]]:
    """[first, last) element ranges of the flat gradient buffer, in the order the backward finalises them.
    graph_layers > 1: unit layers share GAT modules, so no block is final before the backward ends (the library refuses
    layer events there): ONE bucket, the whole buffer, reduced after the backward (`LayerBucketReducer.reduce(flat)`)."""
    if getattr(module, "graph_layers", 1) > 1:
        return [(0, int(module.flat_params().numel()))]
    return [module.layer_param_range(l) for l in block_order(module.layers)]


def _placed_stream(device: torch.device, priority_class: int, rank: int):
    """A stream on the hardware queue that disturbs the current stream and the package's other streams least
    (dvgr_stream_create_placed, csrc/dvgr_place.hip); a plain torch stream where the library is not available (CPU jobs)."""
    try:
        import ctypes as C
        from . import _lib as L_
        lib = L_.load()
    except Exception:
        return torch.cuda.Stream(device=device, priority=-1 if priority_class < 0 else 0)
    h = C.c_void_p()
    with torch.cuda.device(device):
        L_.check(lib.dvgr_stream_create_placed(priority_class, rank, torch.cuda.current_stream(device).cuda_stream, C.byref(h), None, 0),
                 "dvgr_stream_create_placed")
    return torch.cuda.ExternalStream(h.value, device=device)


def _all_reduce_stream_ordered(t: torch.Tensor, op, group) -> None:
    """One collective in the CURRENT stream's order.  NCCL / RCCL: a synchronous-style call (async_op=False) -- the backend
    enqueues the kernel on the current stream itself, where an async_op=True call takes a stream of the process group's
    own (one more hardware queue that spends the step waiting: csrc/dvgr_place.hip).  Other backends: issue, then wait."""
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=op, group=group, async_op=False)
    else:
        dist.all_reduce(t, op=op, group=group, async_op=True).wait()


class CollectiveIssuer:
    """The ONE ordered place from which a process enqueues its gradient collectives on a device.

    RCCL pairs collectives across ranks by issue order on a communicator, so every rank must issue the same sequence.
    The producers are the autograd thread's hooks and the backward's host callbacks: their PROGRAM order is identical on
    every rank (same graph, same bucket tables).  Two reducers with a helper thread each would enqueue in an order that
    depends on thread scheduling -- the head bucket and the unit's MFB block become ready within ~100 us of each other --
    and a mismatch pairs all-reduces of different sizes (hang or corrupt gradients).  So: one FIFO, one helper thread, one
    communication stream per device, shared by `LayerBucketReducer` and `ParamBucketReducer`; items run in posting order."""

    _by_device = {}
    _lock = None

    def __init__(self, device: torch.device):
        import queue
        import threading
        self.device = device
        # high priority, on a hardware queue that shares no command-processor pipe with the caller's stream and the package's
        # other streams (csrc/dvgr_place.hip; an unplaced torch stream here cost the step 2.4 x once the package's streams
        # were placed around it)
        self.stream = _placed_stream(device, priority_class=-1, rank=8)
        self.q = queue.Queue()
        self.errors = []
        self.log = None                      # tests: list that receives a tag per item, in execution order

        def worker():
            torch.cuda.set_device(device)
            while True:
                item = self.q.get()
                try:
                    if item is None:
                        return
                    fn, args, tag = item
                    if self.log is not None:
                        self.log.append(tag)
                    fn(*args)
                except BaseException as e:      # surfaced by the posting reducer's finish()
                    self.errors.append(e)
                finally:
                    self.q.task_done()

        self.thread = threading.Thread(target=worker, daemon=True, name=f"dvgr-collectives-{device}")
        self.thread.start()

    @classmethod
    def get(cls, device: torch.device) -> "CollectiveIssuer":
        import threading
        if cls._lock is None:
            cls._lock = threading.Lock()
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        with cls._lock:
            it = cls._by_device.get(key)
            if it is None:
                it = cls._by_device[key] = CollectiveIssuer(torch.device(*key))
            return it

    def post(self, fn, args=(), tag=None) -> None:
        self.q.put((fn, args, tag))

    def join(self) -> None:
        """Every posted item has been ENQUEUED on the communication stream (not executed: that is stream-ordered)."""
        self.q.join()
        if self.errors:
            err = self.errors[0]
            self.errors.clear()
            raise err


class LayerBucketReducer:
    """Averages a flat gradient buffer across ranks, one all-reduce per layer bucket."""

    def __init__(self, buckets: Sequence[Tuple[int, int]], group=None, device: Optional[torch.device] = None):
        self.buckets = list(buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.stream = None
        self.issuer = None
        if device is not None and device.type == "cuda":
            self.issuer = CollectiveIssuer.get(device)     # shared with every other reducer of this process (issue order)
            self.stream = self.issuer.stream

    def _layer_order(self) -> List[int]:
        """Block index of bucket i (`layer_buckets` order: MFB = L, then L-1 .. 0, then the query block L+1)."""
        return block_order(len(self.buckets) - 2)

    def reduce(self, flat_grad: torch.Tensor, events: Optional[Sequence] = None) -> None:
        """In-place average of `flat_grad`.  `events` is the LAYER-indexed list handed to
        `DualVGRUnit_multiple.set_layer_events` (events[l] = layer block l is final, events[L] = the MFB block, events[L+1]
        = the query block):
        the bucket of layer l is gated on events[l] on the communication stream.  Whatever the events do not cover
        (no list, or a None entry) waits for the caller's current stream instead."""
        if self.world == 1 and not getattr(self, "force", False):
            return
        if events is not None and len(events) != len(self.buckets):
            raise ValueError(f"expected {len(self.buckets)} block-indexed events (L + 2), got {len(events)}")
        order = self._layer_order() if events is not None else [None] * len(self.buckets)
        if flat_grad.is_cuda and self.stream is not None:
            cur = torch.cuda.current_stream(flat_grad.device)
            avg = dist.get_backend(self.group) == "nccl"       # RCCL averages inside the collective (ncclAvg)
            with torch.cuda.stream(self.stream):
                for (a, b), layer in zip(self.buckets, order):
                    ev = events[layer] if events is not None else None
                    if ev is not None:
                        self.stream.wait_event(ev)
                    else:
                        self.stream.wait_stream(cur)
                    _all_reduce_stream_ordered(flat_grad[a:b], dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, self.group)
                    if not avg:
                        flat_grad[a:b].mul_(1.0 / self.world)
            cur.wait_stream(self.stream)
        else:
            for (a, b) in self.buckets:
                dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.mul_(1.0 / self.world)

    # ---- the same reduction, launched bucket by bucket from the backward's host callback --------------------------------
    # (dvgr_opts.layer_done_cb / DualVGRUnit_multiple.set_layer_callback): the collective of a bucket enters the
    # communication stream the moment its event has been enqueued, not after the whole backward has been launched.
    def start(self, num_layers: int, threaded: bool = True) -> None:
        """Begin one step.  threaded: the collectives are enqueued by the process's CollectiveIssuer thread, so that their
        host cost (tens of microseconds each) does not sit in the backward's launch path -- the callback only posts the
        bucket, and the issuer keeps ONE issue order over all reducers."""
        self._works = []
        self._bucket_of_layer = {layer: i for i, layer in enumerate(block_order(num_layers))}
        self._threaded = bool(threaded) and self.issuer is not None

    def _launch_now(self, layer_index: int, flat_grad: torch.Tensor, event=None, caller_stream=None) -> None:
        a, b = self.buckets[self._bucket_of_layer[layer_index]]
        if flat_grad.is_cuda and self.stream is not None:
            with torch.cuda.stream(self.stream):
                if event is not None:
                    self.stream.wait_event(event)
                else:
                    # no event: everything the CALLER's stream (captured by launch(): the helper thread's own current
                    # stream is a different one) has enqueued so far
                    self.stream.wait_stream(caller_stream if caller_stream is not None
                                            else torch.cuda.current_stream(flat_grad.device))
                flat_grad.record_stream(self.stream)           # may be a temporary of the caller: keep it alive for RCCL
                # RCCL averages inside the collective (ncclAvg): no separate scaling pass over the bucket
                avg = dist.get_backend(self.group) == "nccl"
                _all_reduce_stream_ordered(flat_grad[a:b], dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, self.group)
                if not avg:
                    flat_grad[a:b].mul_(1.0 / self.world)
            self._works.append((None, a, b, flat_grad))
        else:
            dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
            flat_grad[a:b].mul_(1.0 / self.world)

    def launch(self, layer_index: int, flat_grad: torch.Tensor, event=None) -> None:
        if self.world == 1 and not getattr(self, "force", False):
            return
        cur = torch.cuda.current_stream(flat_grad.device) if flat_grad.is_cuda else None
        if getattr(self, "_threaded", False) and flat_grad.is_cuda:
            self.issuer.post(self._launch_now, (layer_index, flat_grad, event, cur), ("unit", layer_index))
        else:
            self._launch_now(layer_index, flat_grad, event, cur)

    def finish(self) -> None:
        """After the backward call: scale every launched bucket and make the current stream wait for all of them."""
        if getattr(self, "_threaded", False):
            self.issuer.join()
        works = getattr(self, "_works", [])
        if not works:
            return
        torch.cuda.current_stream(works[0][3].device).wait_stream(self.stream)
        self._works = []


COMM_SYMBOLS = None


def _comm_lib():
    """libdvgr_hip.so with the dvgr_comm_* entry points of include/dvgr_comm.h bound, and RCCL resolved from the copy this
    process already uses (PyTorch's librccl.so): one RCCL per process."""
    import ctypes as C
    import os
    from . import _lib as L_
    global COMM_SYMBOLS
    lib = L_.load()
    if COMM_SYMBOLS is None:
        COMM_SYMBOLS = [
            ("dvgr_comm_load", C.c_int, [C.c_char_p]),
            ("dvgr_comm_unique_id", C.c_int, [C.c_void_p]),
            ("dvgr_comm_create", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
            ("dvgr_comm_destroy", C.c_int, [C.c_void_p]),
            ("dvgr_comm_world", C.c_int, [C.c_void_p]),
            ("dvgr_comm_allreduce_avg", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
            ("dvgr_comm_join", C.c_int, [C.c_void_p, C.c_void_p]),
            ("dvgr_comm_allreduce_avg_on", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
            ("dvgr_comm_place", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_float), C.c_int]),
            ("dvgr_comm_set_inorder", C.c_int, [C.c_void_p, C.c_int]),
            ("dvgr_comm_device_stream", C.c_int, [C.POINTER(C.c_void_p)]),
        ]
        for name, restype, argtypes in COMM_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L_.check(lib.dvgr_comm_load(path.encode() if os.path.exists(path) else None), "dvgr_comm_load")
    return lib


class NativeUnitReducer:
    """The unit's gradient all-reduce without the interpreter in its path (include/dvgr_comm.h): an RCCL communicator of
    its own plus ONE communication stream, created once; `unit.set_comm(self)` makes every backward call enqueue the
    all-reduce (mean over ranks) of each block of the flat gradient buffer right behind the block's own event -- from C,
    inside the call -- in the order the blocks become final.  `finish()` makes the current stream wait.

    The unique id travels over torch.distributed (any backend: it is 128 bytes on the host), so the job's rendezvous stays
    where the launcher put it.  One communicator = one issue order by construction (the backward's program order, the same
    on every rank); it shares nothing with torch's process group, whose collectives (ParamBucketReducer) keep their own."""

    def __init__(self, device: torch.device, group=None, force_single: bool = False, busy_streams: Sequence = (),
                 place: bool = True, inorder: bool = False):
        """inorder: every collective goes onto the stream that produced its buffer (dvgr_comm_set_inorder: no communication
        stream, no cross-stream event; it delays that stream's later work instead of overlapping it).
        busy_streams: torch streams, besides the current one, that carry work during a step (the whole model: its
        question-encoder stream) -- the communication stream is put on a hardware queue that disturbs none of them
        (`place`: dvgr_comm_place, ~0.1 s once; create this object AFTER those streams exist)."""
        import ctypes as C
        from . import _lib as L_
        if device.type != "cuda":
            raise RuntimeError("NativeUnitReducer needs a ROCm device (RCCL); CPU jobs use LayerBucketReducer over gloo")
        self.lib = _comm_lib()
        self.device = device
        if dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        elif force_single:
            self.rank, self.world = 0, 1
        else:
            raise RuntimeError("torch.distributed is not initialised (it carries the RCCL unique id to the ranks)")
        # Rounds 2-3 measured this reducer at 3.3 ms per step instead of 1.1 unless torch's NCCL group had run a collective
        # first.  Round 4 found why (csrc/dvgr_place.hip): the group's streams get their hardware queues at their first use,
        # and queues created AFTER the communication stream was placed can land on a busy stream's pipe.  So the group runs
        # one collective here, before the communicator and its stream exist, and place() measures against everything.
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            warm = torch.zeros(8, device=device, dtype=torch.float32)
            dist.all_reduce(warm, group=group)
            torch.cuda.synchronize(device)
            self.warmed = True
        else:
            self.warmed = False                      # (gloo rendezvous / single process: no other RCCL streams to come)
        # Everything that can fail on ONE rank only -- resolving RCCL (done above), creating the communication stream, the
        # unique id on rank 0 -- happens BEFORE the first collective of the set-up, and the ranks agree on the outcome (MAX of
        # a failure flag) so that they raise together: a rank that failed alone would leave its peers blocked in the id
        # broadcast or in ncclCommInitRank while it moves on to a different collective.
        ident = C.create_string_buffer(128)
        failure = None
        try:
            s_ = C.c_void_p()
            with torch.cuda.device(device):
                L_.check(self.lib.dvgr_comm_device_stream(C.byref(s_)), "dvgr_comm_device_stream")   # creates the stream
            if self.rank == 0:
                L_.check(self.lib.dvgr_comm_unique_id(ident), "dvgr_comm_unique_id")
        except Exception as e:                       # noqa: BLE001 -- reported on every rank below
            failure = e
        if self.world > 1:
            on_dev = dist.get_backend(group) == "nccl"
            flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=device if on_dev else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            if int(flag.item()) and failure is None:
                failure = RuntimeError("NativeUnitReducer: the set-up failed on another rank")
        if failure is not None:
            raise RuntimeError(f"NativeUnitReducer set-up failed (rank {self.rank}): {failure}") from failure
        if self.world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        # ncclCommInitRank and the stream placement come AFTER that first agreement and can still fail on one rank alone (a
        # rank that raised by itself would fall back to another reducer while its peers keep this communicator: mismatched
        # collectives, a hang until the process-group timeout).  So: a failed placement is not an error at all (the library
        # keeps candidate 0 and reports mask -1), and whatever else failed is agreed on a second time before anyone raises.
        failure = None
        self.handle = None
        self.placement = None
        self.inorder = bool(inorder)
        try:
            h = C.c_void_p()
            with torch.cuda.device(device):
                L_.check(self.lib.dvgr_comm_create(ident, self.rank, self.world, C.byref(h)), "dvgr_comm_create")
            self.handle = h.value
            if self.inorder:
                L_.check(self.lib.dvgr_comm_set_inorder(self.handle, 1), "dvgr_comm_set_inorder")
            elif place:
                self.place(busy_streams)
        except Exception as e:                       # noqa: BLE001 -- reported on every rank below
            failure = e
        if self.world > 1:
            on_dev = dist.get_backend(group) == "nccl"
            flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=device if on_dev else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            if int(flag.item()) and failure is None:
                failure = RuntimeError("NativeUnitReducer: creating the communicator failed on another rank")
        if failure is not None:
            self.close()
            raise RuntimeError(f"NativeUnitReducer set-up failed (rank {self.rank}): {failure}") from failure

    def place(self, busy_streams: Sequence = ()) -> dict:
        """dvgr_comm_place against [the current stream] + busy_streams (+ the library's own side / query stream)."""
        import ctypes as C
        from . import _lib as L_
        ptrs = [torch.cuda.current_stream(self.device).cuda_stream] + [s.cuda_stream for s in busy_streams]
        arr = (C.c_void_p * len(ptrs))(*[C.c_void_p(p) for p in ptrs])
        rep = (C.c_float * 64)()
        with torch.cuda.device(self.device):
            L_.check(self.lib.dvgr_comm_place(self.handle, arr, len(ptrs), rep, 64), "dvgr_comm_place")
        self.placement = L_._report(rep)
        return self.placement

    def start(self, num_layers: int = 0) -> None:       # same call shape as LayerBucketReducer
        return None

    def reduce(self, flat_grad: torch.Tensor) -> None:
        """The whole buffer in one collective, behind the current stream (a caller that does not use set_comm)."""
        from . import _lib as L_
        cur = torch.cuda.current_stream(self.device).cuda_stream
        L_.check(self.lib.dvgr_comm_allreduce_avg(self.handle, flat_grad.data_ptr(), flat_grad.numel(), None, cur),
                 "dvgr_comm_allreduce_avg")
        self.finish()

    def finish(self) -> None:
        from . import _lib as L_
        L_.check(self.lib.dvgr_comm_join(self.handle, torch.cuda.current_stream(self.device).cuda_stream), "dvgr_comm_join")

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.dvgr_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeParamReducer:
    """The gradients OUTSIDE the unit (encoders, answer head) on the library's own RCCL communicator as well, so that a
    whole-model data-parallel step has no torch.distributed collective, no issuer thread and no pack / unpack pass in it
    (round-3 review: `bench.py --model --gpus N` depended on the interpreter-driven reducers, +17 % on one rank).

    flat_modules: modules whose backward produces ONE flat gradient buffer (clip.BiLSTM, head.AnswerHead): their backward
        hands that buffer to `grad_sink` the moment it has been enqueued, and it is averaged in place by one collective;
        the parameters' `.grad` are views of it (autograd takes a returned gradient over as it is when `.grad` is None).
    params: every other trainable parameter (embedding table, motion projection): one collective per tensor, from its
        post-accumulate-grad hook.
    All enqueues happen on the autograd thread in the backward's program order -- the same on every rank -- interleaved with
    the unit's own blocks (dvgr_opts.comm) on the ONE communicator.  Every parameter must receive a gradient on every rank
    each step (true for this model: a missing collective on one rank would stall the others), and `.grad` must be None when
    the backward starts (no accumulation over several backward calls: TrainStep clears them)."""

    def __init__(self, native: "NativeUnitReducer", flat_modules: Sequence, params: Sequence[torch.nn.Parameter]):
        self.native = native
        self.flat_modules = list(flat_modules)
        flat_ids = {id(p) for m in self.flat_modules for p in m._plist}
        self.params = [p for p in params if p.requires_grad and id(p) not in flat_ids]
        for m in self.flat_modules:
            m.grad_sink = self._sink
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)
        self._active = False
        self._flats = []
        self.count = 0                   # collectives enqueued in the current / last step

    def _enqueue(self, t: torch.Tensor) -> None:
        from . import _lib as L_
        if not t.is_contiguous():
            raise RuntimeError("NativeParamReducer: a gradient tensor is not contiguous")
        cur = torch.cuda.current_stream(t.device).cuda_stream     # the stream this gradient was produced on (autograd replays a
        L_.check(self.native.lib.dvgr_comm_allreduce_avg(self.native.handle, t.data_ptr(), t.numel(), None, cur),   # node on its forward's)
                 "dvgr_comm_allreduce_avg")
        self.count += 1

    def _sink(self, mod, gflat: torch.Tensor) -> None:
        if not self._active:
            return
        self._enqueue(gflat)
        self._flats.append((mod, gflat))

    def _hook(self, p) -> None:
        if self._active and p.grad is not None:
            self._enqueue(p.grad)

    def start(self) -> None:
        self._active = True
        self._flats = []
        self.count = 0

    def finish(self) -> None:
        self._active = False
        self.native.finish()                                      # the current stream waits for every collective
        for mod, gflat in self._flats:                            # (autograd copied instead of taking the views over: refresh)
            base = gflat.data_ptr()
            for p, off in zip(mod._plist, mod._table):
                if p.grad is not None and p.grad.data_ptr() != base + 4 * off:
                    p.grad.copy_(gflat[off:off + p.numel()].view_as(p.grad))
        self._flats = []


def shard_batch(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [first, last) slice of `n_items` independent QA pairs owned by `rank`."""
    base, rem = divmod(n_items, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


class ParamBucketReducer:
    """Averages the gradients of ordinary nn.Parameters (the encoders and the answer head around the unit) across ranks,
    one all-reduce per bucket, each launched as soon as the LAST gradient of its bucket has been accumulated -- while the
    rest of the backward is still running (what torch's DistributedDataParallel does with its buckets; here on the
    process's CollectiveIssuer -- the same FIFO, helper thread and communication stream as LayerBucketReducer, so that
    every rank issues ONE sequence of collectives -- and without wrapping the module).

    Each bucket reduces a FIXED flat buffer (the numel of all its trainable parameters, zeros where a parameter received
    no gradient this step): the element count of a collective never depends on which gradients a rank happened to get.

    buckets: lists of parameters, ideally in the order the backward finishes them (head first, encoders last).
    Usage per step: start(); loss.backward(); finish()."""

    def __init__(self, buckets: Sequence[Sequence[torch.nn.Parameter]], group=None, device: Optional[torch.device] = None):
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if len(b)]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.stream = None
        self.issuer = None
        if device is not None and device.type == "cuda":
            self.issuer = CollectiveIssuer.get(device)
            self.stream = self.issuer.stream
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)
        self._flat = [None] * len(self.buckets)      # per bucket: the fixed-size buffer the collective runs on
        self._pending = None
        self._active = False
        self._events = []
        self._launched = None
        self.force = False

    def _flat_of(self, bi, like: torch.Tensor) -> torch.Tensor:
        f = self._flat[bi]
        n = sum(p.numel() for p in self.buckets[bi])
        if f is None or f.device != like.device or f.numel() != n:
            f = self._flat[bi] = torch.zeros(n, device=like.device, dtype=torch.float32)
        return f

    # -- autograd thread --------------------------------------------------------------------------------------------------
    def _hook(self, p):
        if not self._active:
            return
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._bucket_ready(bi)

    def _bucket_ready(self, bi):
        """Bucket bi has all its gradients (or finish() declares it complete).  Collectives are POSTED STRICTLY IN BUCKET-INDEX
        ORDER: a ready bucket waits until every lower-index bucket has been posted.  Which gradients a rank received (and
        therefore which buckets complete from a hook and which only in finish()) may differ between ranks; the index order
        does not, so every rank issues the same sequence on the shared CollectiveIssuer."""
        self._ready[bi] = True
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._post_bucket(self._next)
            self._next += 1

    def _post_bucket(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if self.world == 1 and not self.force:
            return
        params = self.buckets[bi]
        like = next((p.grad for p in params if p.grad is not None), params[0].data)
        if like.is_cuda and self.issuer is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(like.device))     # the stream this bucket's last gradient was produced on
            self._events.append(ev)
            self.issuer.post(self._reduce_now, (bi, ev), ("param", bi))
        else:
            self._reduce_cpu(bi)

    def _pack(self, bi, flat):
        o = 0
        for p in self.buckets[bi]:
            n = p.numel()
            if p.grad is not None:
                flat[o:o + n].copy_(p.grad.reshape(-1))
            else:
                flat[o:o + n].zero_()
            o += n

    def _unpack(self, bi, flat):
        o = 0
        for p in self.buckets[bi]:
            n = p.numel()
            red = flat[o:o + n].view_as(p)
            if p.grad is not None:
                p.grad.copy_(red)
            elif self.world > 1:
                p.grad = red.clone()             # another rank had a gradient for it: this rank takes the average as well
            o += n

    # -- issuer thread ----------------------------------------------------------------------------------------------------
    def _reduce_now(self, bi, ev):
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            flat = self._flat_of(bi, self.buckets[bi][0].data)
            self._pack(bi, flat)
            avg = dist.get_backend(self.group) == "nccl"
            _all_reduce_stream_ordered(flat, dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, self.group)
            if not avg:
                flat.mul_(1.0 / self.world)
            self._unpack(bi, flat)
            for p in self.buckets[bi]:
                if p.grad is not None:
                    p.grad.record_stream(self.stream)

    def _reduce_cpu(self, bi):
        flat = self._flat_of(bi, self.buckets[bi][0].data)
        self._pack(bi, flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        self._unpack(bi, flat)

    def start(self) -> None:
        self._pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0                               # lowest bucket index not posted yet
        self._events = []
        self._active = True

    def finish(self) -> None:
        """After backward(): buckets whose parameters did not all receive a gradient are reduced now (in bucket order, the
        same on every rank); then the current stream waits for every collective."""
        self._active = False
        if self.world == 1 and not self.force:
            return
        for bi in range(len(self.buckets)):
            self._bucket_ready(bi)
        if self.issuer is not None:
            self.issuer.join()
            if self._events:
                torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)


def model_param_buckets(model) -> List[List[torch.nn.Parameter]]:
    """Buckets of a DualVGR's non-unit parameters in the order its backward finishes them: answer head; motion Linear +
    appearance encoder; question encoder (embedding included)."""
    head = list(model.feature_aggregation.parameters()) + list(model.output_unit.parameters())
    visual = list(model.visual_motion_input_unit.parameters()) + list(model.visual_appearance_input_unit.parameters())
    ling = list(model.linguistic_input_unit.parameters())
    return [head, visual, ling]
