"""CPU, world_size 2 over gloo: the data-parallel path of the unit (dualvgr_videoqa_amd/parallel.py).

The kernels cannot run here, so each rank computes its shard's gradients with the oracle, lays them
out in the product's flat gradient buffer (the module's parameter table) and runs the product's
LayerBucketReducer; the result must equal the single-process gradient on the concatenated batch
(mean losses average exactly, the batch-SUM HSIC term is scaled by world)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import losses, parallel, synth
from oracle import unit_torch as O

WD, D, N, LR, T, B = 32, 64, 5, 2, 6, 6


def _free_port():
    """A fresh file:// rendezvous (kept under its old name): a TCP port probed as free can be taken again before the store
    binds it."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="dvgr_pg_")
    os.close(fd)
    os.unlink(path)              # the FileStore creates it
    return "file://" + path


def _grads(sd_np, inp, lo, hi, world):
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
    t = {k: torch.from_numpy(v[lo:hi].copy()) for k, v in inp.items()}
    outs = O.unit_forward(sd, t["A"], t["M"], t["dq"], t["words"], t["qlen"], LR, 1, naive=False)
    # per-rank objective: means over the local shard, HSIC (a batch sum) times world
    loss = outs[0].mean() + losses.unit_aux_losses(outs[3], outs[4], outs[5], outs[6], N)[0] \
        + 1e-3 * world * losses.unit_aux_losses(outs[3], outs[4], outs[5], outs[6], N)[1]
    loss.backward()
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in sd.items()}


def _worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, LR), seed=5, bias_range=0.1)
        inp = synth.unit_inputs(B, N, T, D, WD, seed=5)
        lo, hi = parallel.shard_batch(B, rank, world)
        g = _grads(sd_np, inp, lo, hi, world)
        m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=LR)
        flat = torch.zeros(m._flat.numel())
        for (name, shape, off) in m._table:
            flat[off:off + g[name].numel()] = g[name].reshape(-1)
        buckets = parallel.layer_buckets(m)
        assert buckets[0] == m.layer_param_range(LR) and buckets[-2] == m.layer_param_range(0) and buckets[-1] == m.layer_param_range(LR + 1)
        assert sorted(buckets)[0][0] == 0 and sorted(buckets)[-1][1] == flat.numel()
        flat2 = flat.clone()
        one_shot = parallel.LayerBucketReducer(buckets)
        assert one_shot._layer_order() == [LR] + list(range(LR - 1, -1, -1)) + [LR + 1]    # bucket i <-> block-indexed event
        try:
            one_shot.reduce(flat, [None] * LR)                                   # L + 2 block-indexed events expected
            raise AssertionError("a short event list must be refused")
        except ValueError:
            pass
        one_shot.reduce(flat, [None] * (LR + 2))
        # the same reduction driven bucket by bucket, in the order the backward's host callback fires (L, L-1, .., 0)
        red = parallel.LayerBucketReducer(buckets)
        red.start(LR)
        for layer in parallel.block_order(LR):
            red.launch(layer, flat2)
        red.finish()
        assert torch.equal(flat, flat2)
        if rank == 0:
            q.put(flat.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_average_equals_full_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process, whole batch, same objective with world = 1
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, LR), seed=5, bias_range=0.1)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=5)
    g = _grads(sd_np, inp, 0, B, 1)
    m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=LR)
    for (name, shape, off) in m._table:
        want = g[name].reshape(-1).numpy()
        got = flat[off:off + want.size]
        scale = max(float(np.abs(want).max()), 1e-3)
        assert float(np.abs(got - want).max()) < 2e-4 * scale, name


def _worker_params(rank, world, port, q):
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
        unused = torch.nn.Parameter(torch.ones(4))                  # a bucket member that never gets a gradient
        buckets = [list(net[3].parameters()), list(net[2].parameters()) + [unused], list(net[0].parameters())]
        red = parallel.ParamBucketReducer(buckets)
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10 + rank
        for _ in range(2):                                          # two steps: the hooks re-arm
            net.zero_grad(set_to_none=True)
            red.start()
            net(x).pow(2).mean().backward()
            red.finish()
        if rank == 0:
            q.put([p.grad.numpy().copy() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_param_bucket_reducer_averages_ordinary_parameters():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_params, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    want = [torch.zeros_like(p) for p in net.parameters()]
    for rank in range(world):
        net.zero_grad(set_to_none=True)
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10 + rank
        net(x).pow(2).mean().backward()
        for w, p in zip(want, net.parameters()):
            w += p.grad / world
    for g, w in zip(got, want):
        assert np.allclose(g, w.numpy(), rtol=1e-6, atol=1e-7)


def _worker_order(rank, world, port, q):
    """Gradient presence differs between the ranks: on rank 1 one parameter of bucket 0 receives no gradient, so bucket 0
    completes from a hook on rank 0 and only in finish() on rank 1, while buckets 1 and 2 complete from hooks on both."""
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        a = torch.nn.Linear(4, 3)
        b = torch.nn.Linear(3, 2)
        side = torch.nn.Parameter(torch.ones(3))                    # bucket 0 member, used on rank 0 only
        c = torch.nn.Linear(2, 2)
        buckets = [list(c.parameters()) + [side], list(b.parameters()), list(a.parameters())]   # backward order: c, b, a
        red = parallel.ParamBucketReducer(buckets)
        order = []
        real_cpu = red._reduce_cpu
        red._reduce_cpu = lambda bi: (order.append(bi), real_cpu(bi))[1]
        x = torch.arange(8, dtype=torch.float32).reshape(2, 4) / 5 + rank
        red.start()
        h = b(torch.tanh(a(x)))
        y = c(h) + ((h[:, :1] * side[:1]).sum() if rank == 0 else 0.0)
        y.pow(2).mean().backward()
        red.finish()
        q.put((rank, order, [None if p.grad is None else p.grad.numpy().copy() for bk in buckets for p in bk]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_param_buckets_are_issued_in_index_order_whatever_gradients_a_rank_received():
    """the size of each collective was rank-invariant but the ISSUE ORDER followed which buckets completed
    from a hook.  Every rank must post bucket 0, 1, 2 in that order; a parameter without a local gradient takes the average."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_order, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, order, grads = q.get(timeout=240)
        res[r] = (order, grads)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == [0, 1, 2], (res[0][0], res[1][0])
    for g0, g1 in zip(res[0][1], res[1][1]):
        assert g0 is not None and g1 is not None
        assert np.allclose(g0, g1, rtol=1e-6, atol=1e-7)            # both ranks hold the same averaged gradients


def test_shard_batch_partitions():
    for n in (1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_batch(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
