#!/usr/bin/env python3
"""Where the HOST time of the whole-model training step (bench.py --model) goes: every C-ABI entry point timed (wall clock
inside the call = launches + event calls it makes), the rest is Python / autograd / torch operators around them.
    python tools/model_host_profile.py [steps]"""
import collections
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L_, synth

B, N, Lr, T, D, WD = 64, 20, 3, 15, 768, 300
V, FR, VQ, VA = 2048, 16, 4000, 1000
dev = torch.device("cuda", 0)
vocab = {"question_token_to_idx": {str(i): i for i in range(VQ)}, "answer_token_to_idx": {str(i): i for i in range(VA)}}
torch.manual_seed(666)
m = dv.DualVGR(vision_dim=V, module_dim=D, word_dim=WD, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1,
               unit_layers=Lr, dtype="bf16x3").to(dev).train()
m.visual_input_unit.reuse_buffers = True
app = torch.from_numpy(np.abs(synth.normal("b_app", (B, N, FR, V), 0.5, 666))).to(dev)
mot = torch.from_numpy(np.abs(synth.normal("b_mot", (B, N, V), 0.5, 666))).to(dev)
qlen = torch.from_numpy((3 + (synth.uniform("b_ql", (B,), 0.0, 1.0, 666) * (T - 2)).astype(np.int64)).clip(1, T)).to(dev)
qlen[0] = T
q = torch.from_numpy((2 + (synth.uniform("b_q", (B, T), 0.0, 1.0, 666) * (VQ - 2)).astype(np.int64)).clip(2, VQ - 1)).to(dev)
q = q * (torch.arange(T, device=dev)[None, :] < qlen[:, None])
answers = torch.from_numpy((synth.uniform("b_ans", (B,), 0.0, float(VA), 666)).astype(np.int64)).to(dev)
unit_red = param_red = None
_pads = []
for spec in [x for x in os.environ.get("MHP_PAD", "").split(",") if x]:      # e.g. "h2,n1": throwaway streams of a priority, each
    pr = {"h": -1, "n": 0, "l": 1}[spec[0]]                                   # used once, created BEFORE the reducer's streams
    for _ in range(int(spec[1:])):                                            # (shifts which hardware queue / pipe the later ones get)
        s_ = torch.cuda.Stream(device=dev, priority=pr)
        with torch.cuda.stream(s_):
            torch.zeros(8, device=dev).add_(1.0)
        _pads.append(s_)
torch.cuda.synchronize()
MHP = os.environ.get("MHP_NATIVE", "0")               # 1 = both reducers, unit = only the unit's in-call communicator, param = only
if MHP in ("1", "unit", "param", "dry", "inorder", "pgonly", "commonly", "gloo"):               # the encoders' / head's collectives, dry = those as event edges without RCCL      # the native reducers on ONE rank (bench.py's DVGR_DDP_SELFTEST set-up)
    import torch.distributed as dist
    from dualvgr_videoqa_amd import parallel
    from dualvgr_videoqa_amd.clip import BiLSTM
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1,
                            pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("MHP_PG_HIGH", "1") == "1"))
    if MHP == "pgonly":                 # torch's NCCL group initialised and used once, nothing else
        w_ = torch.zeros(8, device=dev)
        dist.all_reduce(w_)
        torch.cuda.synchronize()
    do_place = os.environ.get("MHP_PLACE", "1") == "1"
    if MHP != "pgonly":
        unit_red = parallel.NativeUnitReducer(dev, force_single=True, place=do_place,
                                              busy_streams=[L_.streams(dev)["encoder"]] if do_place else ())
        print("placement:", unit_red.placement)
    if MHP in ("1", "unit"):
        m.visual_input_unit.set_comm(unit_red)
    flat_mods = [m._head] + [x for x in m.modules() if isinstance(x, BiLSTM)]
    unit_ids = {id(p) for p in m.visual_input_unit.parameters()}
    if MHP in ("1", "param", "dry", "inorder"):
        param_red = parallel.NativeParamReducer(unit_red, flat_mods, [p for p in m.parameters() if id(p) not in unit_ids])
    if MHP == "inorder":
        def inorder(t):
            cur = torch.cuda.current_stream(dev).cuda_stream
            L_.check(unit_red.lib.dvgr_comm_allreduce_avg_on(unit_red.handle, t.data_ptr(), t.numel(), cur), "allreduce_on")
            param_red.count += 1
        param_red._enqueue = inorder
        param_red.native = type("NoJoin", (), {"finish": staticmethod(lambda: None)})()
    if MHP == "dry":
        import ctypes as C_
        h_ = C_.c_void_p()
        L_.check(unit_red.lib.dvgr_comm_device_stream(C_.byref(h_)), "stream")
        cs_ = torch.cuda.ExternalStream(h_.value, device=dev)

        DRY = os.environ.get("MHP_DRY", "both")        # both | waits (nobody waits for the stream) | join (it waits for nothing) | torch
        if DRY == "torch":                              # (an ordinary torch stream in the communication stream's role)
            cs_ = torch.cuda.Stream(device=dev)

        def dry(t):
            if DRY != "join":
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                cs_.wait_event(ev)
            param_red.count += 1
        param_red._enqueue = dry

        def dry_finish():
            if DRY != "waits":
                ev = torch.cuda.Event()
                ev.record(cs_)
                torch.cuda.current_stream(dev).wait_event(ev)
        param_red.native = type("DryJoin", (), {"finish": staticmethod(dry_finish)})()
m.overlap_encoders = os.environ.get("DVGR_NO_ENCODER_OVERLAP", "0") != "1"
ts = dv.TrainStep(m, lr=1e-4, alpha=1.0, beta=1e-8, max_norm=12.0, unit_reducer=unit_red if MHP in ("1", "unit") else None,
                  param_reducer=param_red)
if MHP in ("param", "dry", "inorder"):
    _pf = param_red.finish
    param_red.finish = lambda: _pf()      # (its finish() joins the communication stream by itself)

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


class Timed:
    def __init__(self, fn, key):
        self.fn, self.key = fn, key

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        acc[self.key] += time.perf_counter() - t0
        cnt[self.key] += 1
        return r


def wrap(lib):
    for name in dir(lib):
        if name.startswith("dvgr_"):
            try:
                setattr(lib, name, Timed(getattr(lib, name), name))
            except Exception:
                pass


def step():
    ts(app, mot, q, qlen, answers)


if param_red is not None:                                # the reducers' Python entry points, timed like the C ones
    param_red._enqueue = Timed(param_red._enqueue, "py:NativeParamReducer._enqueue")
    param_red.finish = Timed(param_red.finish, "py:NativeParamReducer.finish")
_bw = torch.autograd.backward
torch.autograd.backward = Timed(_bw, "py:torch.autograd.backward (whole)")


for _ in range(8):
    step()
torch.cuda.synchronize()
libs = [L_.load()]
for lib in libs:
    wrap(lib)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
launch0 = None
for lib in libs:
    if hasattr(lib, "dvgr_dbg_launch_count"):
        launch0 = lib
c0 = launch0.dvgr_dbg_launch_count.fn() if launch0 is not None else 0
# mode "sync" (argv[2]): synchronise before every step, so no call ever blocks on a full queue -- the pure enqueue cost
sync_each = len(sys.argv) > 2 and sys.argv[2] == "sync"
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for _ in range(n):
    if sync_each:
        torch.cuda.synchronize()
    h0 = time.perf_counter()
    step()
    host += time.perf_counter() - h0
t1 = t0 + host
torch.cuda.synchronize()
t2 = time.perf_counter()
c1 = launch0.dvgr_dbg_launch_count.fn() if launch0 is not None else 0
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, device-complete {1e3 * (t2 - t0) / n:.3f} ms/step, "
      f"library launches per step {(c1 - c0) / n:.1f}")
tot = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    if cnt[k] == 0:
        continue
    print(f"  {k:36s} {cnt[k] / n:6.1f} calls/step  {1e3 * v / n:7.3f} ms/step  {1e6 * v / cnt[k]:8.1f} us/call")
    tot += v
print(f"  inside C-ABI calls: {1e3 * tot / n:.3f} ms/step; outside (Python, autograd, torch ops): {1e3 * ((t1 - t0) - tot) / n:.3f} ms/step")
