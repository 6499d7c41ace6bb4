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
ts = dv.TrainStep(m, lr=1e-4, alpha=1.0, beta=1e-8, max_norm=12.0)

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
