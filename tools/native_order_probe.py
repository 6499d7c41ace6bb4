#!/usr/bin/env python3
"""Probe: which set-up order of the native reducer gives the fast step.  usage: native_order_probe.py <backend> <plain_first 0|1>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
import torch, torch.distributed as dist
from dualvgr_videoqa_amd import parallel
backend, plain_first = sys.argv[1], int(sys.argv[2])
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
elif backend == "nccl_hi":
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
elif backend == "nccl_hi_barrier":
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
    dist.barrier()
elif backend == "nccl_barrier":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
elif backend == "nccl_lazy_barrier":
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    dist.barrier()
elif backend == "nccl_hi_allreduce":
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
    x = torch.ones(8, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
elif backend == "nccl_lazy":
    dist.init_process_group("nccl", rank=0, world_size=1)
else:
    dist.init_process_group("gloo", rank=0, world_size=1)
red = parallel.NativeUnitReducer(torch.device("cuda", 0)) if os.environ.get("NO_NATIVE") != "1" else None
import test_hip_parallel as T
m, t = T._module()
m.reuse_buffers = True
def run(reducer, steps=60):
    def one():
        m.zero_grad_flat(); T._step(m, t)
        if reducer is not None: reducer.finish()
    for _ in range(15): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
if plain_first:
    print(f"{backend} plain first: plain {run(None):.3f} ms")
if red is not None:
    m.set_comm(red)
    print(f"{backend} plain_first={plain_first}: native {run(red):.3f} ms")
    m.set_comm(None)
print(f"{backend} plain after: {run(None):.3f} ms")
dist.destroy_process_group()
