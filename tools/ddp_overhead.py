#!/usr/bin/env python3
"""Where the data-parallel plumbing's single-rank overhead goes (C2, bf16x3, training): the plain step against the step
with (a) layer events only (the library then folds its query stream into the side stream), (b) events + a no-op host
callback, (c) events + callback posting to the issuer thread without a collective, (d) the full reducer over RCCL."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch
import torch.distributed as dist
import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import parallel, synth

B, N, Lr, T, D, WD = 64, 20, 3, 15, 768, 300
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(9)
m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")   # (its own random init)
m = m.cuda().train()
m.loss_in_forward = (1.0, 1e-8, 1.0)
m.reuse_buffers = True
inp = synth.unit_inputs(B, N, T, D, WD, seed=9)
t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
for k in ("A", "M", "dq", "words"):
    t[k].requires_grad_(True)


def run(pre=None, post=None, steps=200):
    def one():
        m.zero_grad_flat()
        if pre:
            pre()
        for k in ("A", "M", "dq", "words"):
            t[k].grad = None
        raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        _, g = m.last_loss
        torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])
        if post:
            post()
    for _ in range(20):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print(f"plain                          {run():.4f} ms")
events = [torch.cuda.Event() for _ in range(Lr + 2)]
for e in events:
    e.record()
m.set_layer_events(events)
print(f"+ layer events                 {run():.4f} ms")
m.set_layer_callback(lambda idx, g: None)
print(f"+ no-op python callback        {run():.4f} ms")
red = parallel.LayerBucketReducer(parallel.layer_buckets(m), device=torch.device("cuda", 0))
red.force = True
m.set_layer_callback(lambda idx, g: red.issuer.post(lambda: None, (), ("noop", idx)))
print(f"+ post to the issuer thread    {run(post=red.issuer.join):.4f} ms")
m.set_layer_callback(lambda idx, g: red.launch(idx, g, events[idx]))
print(f"+ all-reduce per bucket (RCCL) {run(pre=lambda: red.start(Lr), post=red.finish):.4f} ms")
dist.destroy_process_group()
