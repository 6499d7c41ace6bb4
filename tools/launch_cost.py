#!/usr/bin/env python3
"""Host cost of the primitives the schedules are made of: one kernel launch through the C ABI, an event fork/join between
the caller's stream and a second stream, a small memset.  GPU only."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualvgr_videoqa_amd import _lib as L

lib = L.load()
x = torch.empty(1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
n = 2000


def timeit(name, fn, per=1):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:58s} host {1e6 * (t1 - t0) / n / per:6.2f} us   host+drain {1e6 * (t2 - t0) / n / per:6.2f} us")


timeit("C-ABI launch of a trivial kernel (dvgr_dbg_keep_mask)", lambda: lib.dvgr_dbg_keep_mask(1, 1024, x.data_ptr(), st))
timeit("torch trivial kernel (x.add_(1))", lambda: x.add_(1.0))
timeit("torch memset (x.zero_())", lambda: x.zero_())
s2 = torch.cuda.Stream()
cur = torch.cuda.current_stream()
ev = [torch.cuda.Event() for _ in range(2)]


def forkjoin():
    ev[0].record(cur); s2.wait_event(ev[0])
    lib.dvgr_dbg_keep_mask(1, 1024, x.data_ptr(), s2.cuda_stream)
    ev[1].record(s2); cur.wait_event(ev[1])
    lib.dvgr_dbg_keep_mask(1, 1024, x.data_ptr(), st)


timeit("fork + launch on 2nd stream + join + launch (torch events)", forkjoin)
big = torch.empty(14_800_000, device="cuda")
timeit("hipMemsetAsync 59 MB (big.zero_())", lambda: big.zero_())
A = torch.randn(1280, 768, device="cuda")
Bm = torch.randn(1536, 768, device="cuda")
Cm = torch.empty(1280, 1536, device="cuda")
timeit("C-ABI contraction launch 1280x1536x768 (fp32 ring)", lambda: lib.dvgr_dbg_gemm(0, 1280, 1536, 768, 0, A.data_ptr(), None, 768, Bm.data_ptr(), None, 768,
                                                                                      Cm.data_ptr(), 1536, None, 0, 0, 0, 0, 0, 0, st))
