#!/usr/bin/env python3
"""What RCCL's INFO log says about a one-rank communicator and an all-reduce (the lines bench.py's rccl_info_from_log parses).
usage: python tools/rccl_log_probe.py > out.txt"""
import os, sys, glob
log = "/tmp/dvgr_rccl_probe_%p.log"
os.environ["NCCL_DEBUG"] = "INFO"
os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,TUNING,COLL"
os.environ["NCCL_DEBUG_FILE"] = log
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(3551360, device=dev)
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
dist.destroy_process_group()
for fn in glob.glob(log.replace("%p", "*")):
    lines = open(fn, errors="replace").read().splitlines()
    print("==", fn, len(lines), "lines")
    for l in lines:
        if any(k in l for k in ("hannel", "Algo", "AllReduce", "Init COMPLETE", "nranks", "Ring", "Tree")):
            print(l[:220])
