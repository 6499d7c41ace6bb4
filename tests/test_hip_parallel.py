"""GPU: the hooks the data-parallel path hangs on (dvgr_opts.layer_done_events / layer_done_cb, parallel.LayerBucketReducer)
on one device: an event must not fire before EVERY gradient of its layer block is final (the block is written on two
streams), the reducer must gate each bucket on its own layer's event, and accumulation into a live buffer must refuse
to combine with asynchronous reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import parallel, synth
from oracle import unit_torch as O

pytestmark = pytest.mark.gpu
B, N, Lr, T, D, WD = 64, 20, 3, 15, 768, 300


def _module(train=True):
    sd = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=9, bias_range=0.1)
    m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.cuda().train(train)
    m.loss_in_forward = (1.0, 1e-3, 1.0)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=9)
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    return m, t


def _step(m, t):
    for k in ("A", "M", "dq", "words"):
        t[k].grad = None
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    _, g = m.last_loss
    torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])


def test_layer_event_fires_only_when_the_whole_layer_block_is_final():
    """A third stream waits on event[l] alone, snapshots layer l's slice of the flat gradient, and the snapshot must equal
    the final gradient bit for bit -- also for the tensors written on the backward's MAIN stream (dWcat) while the event is
    recorded on the side stream.  Several steps back to back, training mode, full C2 size."""
    m, t = _module()
    events = [torch.cuda.Event() for _ in range(Lr + 2)]
    m.set_layer_events(events)
    third = torch.cuda.Stream()
    snaps = {}

    def cb(idx, gflat):
        a, b = m.layer_param_range(idx)
        with torch.cuda.stream(third):
            third.wait_event(events[idx])
            snaps[idx] = gflat[a:b].clone()

    m.set_layer_callback(cb)
    for rep in range(6):
        snaps.clear()
        m.zero_grad_flat()
        _step(m, t)
        torch.cuda.synchronize()
        assert sorted(snaps) == list(range(Lr + 2))
        for idx, snap in snaps.items():
            a, b = m.layer_param_range(idx)
            assert torch.equal(snap, m.last_flat_grad[a:b]), (rep, idx)
            assert float(snap.abs().max()) > 0


def test_accumulation_with_the_layer_callback_is_refused():
    m, t = _module(train=False)
    events = [torch.cuda.Event() for _ in range(Lr + 2)]
    m.set_layer_events(events)
    m.set_layer_callback(lambda idx, g: None)
    m.zero_grad_flat()
    _step(m, t)                      # installs the flat buffer
    with pytest.raises(RuntimeError, match="zero_grad_flat"):
        _step(m, t)                  # no zero_grad_flat(): would accumulate a temporary that the reducer owns
    m.set_layer_callback(None)
    _step(m, t)                      # without the callback accumulation is fine
    torch.cuda.synchronize()


def _rendezvous():
    """A fresh file:// rendezvous (no TCP port: a port probed as free can be taken again before the store binds it --
    EADDRINUSE once in a while under the full suite)."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="dvgr_pg_")
    os.close(fd)
    os.unlink(path)              # the FileStore creates it
    return "file://" + path


def test_reducer_gates_each_bucket_on_its_own_layers_event():
    """One rank over RCCL (the collectives really run): reduce() takes the LAYER-indexed event list of set_layer_events and
    must wait for events[L], events[L-1], ..., events[0] in its bucket order; the averaged buffer equals the input."""
    dist.init_process_group("nccl", init_method=_rendezvous(), rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m, t = _module(train=False)
        events = [torch.cuda.Event() for _ in range(Lr + 2)]
        m.set_layer_events(events)
        m.zero_grad_flat()
        _step(m, t)
        want = m.last_flat_grad.clone()
        red = parallel.LayerBucketReducer(parallel.layer_buckets(m), device=torch.device("cuda", 0))
        red.force = True
        waited = []
        real_wait = red.stream.wait_event
        red.stream.wait_event = lambda ev: (waited.append(ev), real_wait(ev))[1]
        got = m.last_flat_grad.clone()
        red.reduce(got, events)
        torch.cuda.synchronize()
        assert [id(e) for e in waited] == [id(events[l]) for l in parallel.block_order(Lr)]
        assert torch.equal(got, want)
        with pytest.raises(ValueError):
            red.reduce(got, events[:-1])
        # bucket-by-bucket form without events: waits for the CALLER's stream (captured in launch(), not the helper thread's)
        got2 = m.last_flat_grad.clone()
        red.start(Lr)
        for layer in parallel.block_order(Lr):
            red.launch(layer, got2)
        red.finish()
        torch.cuda.synchronize()
        assert torch.equal(got2, want)
    finally:
        dist.destroy_process_group()


def _stall_child(q):
    import time
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=_rendezvous(), rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # the reducer (and with it the communication stream) BEFORE the unit's first call: the runtime maps streams to hardware
        # queues in creation order, and a reducer created after the unit had run measured 2.4 ms per step instead of 1.25
        red = parallel.LayerBucketReducer([(0, 1)], device=torch.device("cuda", 0))
        m, t = _module()
        m.reuse_buffers = True
        red.buckets = parallel.layer_buckets(m)
        red.force = True

        def run(reducer, steps=40):
            def one():
                m.zero_grad_flat()
                if reducer is not None:
                    reducer.start(Lr)
                _step(m, t)
                if reducer is not None:
                    reducer.finish()
            for _ in range(10):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps

        plain = run(None)
        events = [torch.cuda.Event() for _ in range(Lr + 2)]
        m.set_layer_events(events)
        m.set_layer_callback(lambda idx, g: red.launch(idx, g, events[idx]))
        q.put(("ok", plain, run(red)))
    except BaseException as e:
        q.put(("error", repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_the_reducers_streams_do_not_stall_the_step():
    """Guard for a regression seen in round 2: with one more internal stream in the backward, a data-parallel step (helper
    thread + reducer stream + RCCL beside the call) took 2.6 x the time of the plain step -- the time went to the host
    between launches.  The module drops its query stream whenever layer events are installed; here the step with the
    torch.distributed reducer running (one rank over RCCL, collectives forced; fresh process, reducer created before the
    unit's first call as bench.py does) must stay well below that regression.  The ratio depends on the HOST (this path is bound
    by the interpreter's enqueue rate: 1.17-1.2 x on the round-2/3 boxes, 1.55 x on a slower round-4 host), so the gate is the
    regression's signature (>= 2 x), not a performance target -- bench.py reports the actual overhead."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_stall_child, args=(q,))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert res[0] == "ok" and p.exitcode == 0, res
    _, plain, with_reducer = res
    print(f"  plain {plain * 1e3:.3f} ms, with the reducer {with_reducer * 1e3:.3f} ms per step")
    assert with_reducer < 2.0 * plain, (plain, with_reducer)


# ---- two ranks, BOTH reducers, one issue order ------------------------------------------------------------------------
def _ddp_worker(rank, world, rendezvous, q, G=1):
    """One data-parallel rank of TrainStep on GPU 0 (the box has one GPU: gloo carries the CUDA tensors, RCCL would refuse
    two ranks on one device).  Runs (a) two steps with the threaded reducers and (b) the same two steps with gradients
    averaged synchronously after each backward, and reports parameters of both plus the issuer's log."""
    import numpy as np
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method=rendezvous, rank=rank, world_size=world)
    try:
        from test_hip_model import model_inputs
        from dualvgr_videoqa_amd import trainer
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        V, D_, Wd, N_, Fr, L_, B_, T_, vq, va, seed = 32, 64, 32, 8, 4, 2, 8, 7, 30, 12, 77
        vocab = {"question_token_to_idx": {str(i): i for i in range(vq)}, "answer_token_to_idx": {str(i): i for i in range(va)}}

        def make():
            torch.manual_seed(1234)              # same initial weights on every rank (as after a broadcast)
            m = dv.DualVGR(vision_dim=V, module_dim=D_, word_dim=Wd, vocab=vocab, num_of_nodes=N_, graph_module="GAT",
                           graph_layers=G, unit_layers=L_, dtype="fp32").to(dev).train()
            m.set_dropout_enabled(False)
            return m

        app, mot, qq, qlen = [torch.from_numpy(x) for x in model_inputs(B_, N_, Fr, V, T_, vq, seed)]
        ans = torch.from_numpy((synth.uniform("ddp_ans", (B_,), 0.0, float(va), seed)).astype(np.int64))
        lo, hi = parallel.shard_batch(B_, rank, world)
        batch = [x[lo:hi].to(dev) for x in (app, mot, qq, qlen, ans)]

        # (a) the product path: both reducers on the shared issuer
        m = make()
        unit = m.visual_input_unit
        events = [torch.cuda.Event() for _ in range(L_ + 2)]
        for ev in events:
            ev.record()
        unit.set_layer_events(events)
        ured = parallel.LayerBucketReducer(parallel.layer_buckets(unit), device=dev)
        pred = parallel.ParamBucketReducer(parallel.model_param_buckets(m), device=dev)
        assert ured.issuer is pred.issuer and ured.stream is pred.stream
        unit.set_layer_callback(lambda idx, g: ured.launch(idx, g, events[idx]))
        log = ured.issuer.log = []
        ts = dv.TrainStep(m, lr=1e-3, alpha=1.0, beta=1e-3, max_norm=1e9, world=world, unit_reducer=ured, param_reducer=pred)
        def grads_of(model):
            return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).float()
                              for p in model.parameters()]).cpu()

        ts(*batch)
        torch.cuda.synchronize()
        got_g = grads_of(m)                      # the AVERAGED gradients of step 1 (max_norm is huge: written back unclipped)
        ts(*batch)
        torch.cuda.synchronize()
        got = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()

        # (b) reference: local backward, then ONE synchronous average of every gradient, then the same optimiser
        m2 = make()
        ts2 = dv.TrainStep(m2, lr=1e-3, alpha=1.0, beta=1e-3, max_norm=1e9, world=world)
        real_step = ts2.opt.step

        def averaged_step(segs):
            torch.cuda.synchronize()
            for seg in segs:
                g = seg[2]
                h = g.cpu()
                dist.all_reduce(h)
                g.copy_((h / world).to(dev))
            return real_step(segs)

        ts2.opt.step = averaged_step
        ts2(*batch)
        torch.cuda.synchronize()
        want_g = grads_of(m2)
        q.put((rank, got.numpy(), got_g.numpy(), want_g.numpy(), list(log)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_with_both_reducers_issue_one_order_and_average_exactly():
    """LayerBucketReducer and ParamBucketReducer used a helper thread EACH on the same process
    group, so the order of their collectives raced.  Both now post to one CollectiveIssuer.  Two gloo ranks on this GPU run
    TrainStep with both reducers: the issue sequence is identical on both ranks and interleaves the two reducers, both
    ranks end with identical parameters, and those equal a step whose gradients were averaged synchronously."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rendezvous = _rendezvous()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, rendezvous, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, got, got_g, want_g, log = q.get(timeout=540)
        res[r] = (got, got_g, want_g, log)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][3] == res[1][3], (res[0][3], res[1][3])                     # one issue order on every rank
    kinds = [t[0] for t in res[0][3]]
    assert "unit" in kinds and "param" in kinds
    first_param, last_unit = kinds.index("param"), len(kinds) - 1 - kinds[::-1].index("unit")
    assert first_param < last_unit                                            # the two reducers really interleave
    import numpy as np
    assert np.array_equal(res[0][0], res[1][0])                               # replicas stay identical
    # == synchronous averaging: the averaged gradients of the first step, per rank, against gradients averaged by one
    # blocking all-reduce after the backward (the backward's float atomics reorder between the two runs: 2e-5 of the largest)
    for r in range(world):
        scale = float(np.abs(res[r][2]).max())
        assert float(np.abs(res[r][1] - res[r][2]).max()) < 2e-5 * scale, r


# ---- the native path: all-reduces enqueued by the backward call itself (include/dvgr_comm.h) ----------------------------
# Each native test body runs in a fresh child process (a process has ONE communication stream per device, placed once).
def _native_child(which, q):
    import time
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # torch's NCCL group, initialised eagerly: it carries the unique id, and its first collective (run by the reducer's
    # constructor) is what keeps the step fast (parallel.NativeUnitReducer)
    dist.init_process_group("nccl", init_method=_rendezvous(), rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        if which == "late":
            # everything the rounds 2-3 rule forbade: the unit has run (side + query stream exist), two more streams carry
            # work (six hardware queues on the command processor's four pipes), THEN the communicator is created --
            # dvgr_comm_place must find it a queue that disturbs neither the caller's stream nor the side stream
            m, t = _module()
            m.reuse_buffers = True
            extra = [torch.cuda.Stream(), torch.cuda.Stream()]
            for s_ in extra:
                with torch.cuda.stream(s_):
                    torch.zeros(64, device="cuda").add_(1.0)
            for _ in range(3):
                m.zero_grad_flat()
                _step(m, t)
            torch.cuda.synchronize()
        red = parallel.NativeUnitReducer(torch.device("cuda", 0), inorder=which.startswith("inorder"))
        assert red.world == 1 and red.handle and red.warmed
        if which == "inorder_correct":
            # the in-order form: each block's collective on the stream that produced it, no communication stream, finish() a no-op
            m, t = _module(train=False)
            m.zero_grad_flat()
            _step(m, t)
            torch.cuda.synchronize()
            want = m.last_flat_grad.clone()
            m.set_comm(red)
            for rep in range(4):
                m.zero_grad_flat()
                _step(m, t)
                red.finish()
                torch.cuda.synchronize()
                assert float((m.last_flat_grad - want).abs().max()) <= 2e-6 * float(want.abs().max()), rep
            m.set_comm(None)
            buf = want.clone()
            red.reduce(buf)
            torch.cuda.synchronize()
            assert torch.equal(buf, want) and red.placement is None
            q.put(("ok", None))
        elif which == "correct":
            m, t = _module(train=False)
            m.zero_grad_flat()
            _step(m, t)
            torch.cuda.synchronize()
            want = m.last_flat_grad.clone()
            m.set_comm(red)
            for rep in range(4):
                m.zero_grad_flat()
                _step(m, t)
                red.finish()
                torch.cuda.synchronize()
                # (two backward runs: their float atomics add in different orders)
                assert float((m.last_flat_grad - want).abs().max()) <= 2e-6 * float(want.abs().max()), rep
            try:
                _step(m, t)                                        # would accumulate into a buffer the collectives own
                raise AssertionError("accumulation beside the native reducer must be refused")
            except RuntimeError as e:
                assert "zero_grad_flat" in str(e)
            m.set_comm(None)
            buf = want.clone()                                     # stand-alone form: one collective over a whole buffer
            red.reduce(buf)
            torch.cuda.synchronize()
            assert torch.equal(buf, want)
            # a second communicator of the process shares the device's (already placed) communication stream
            second = parallel.NativeUnitReducer(torch.device("cuda", 0))
            assert second.placement["candidates"] == 0
            second.close()
            q.put(("ok", red.placement))
        else:
            if which != "late":
                m, t = _module()
                m.reuse_buffers = True

            def run(reducer, steps=60):
                def one():
                    m.zero_grad_flat()
                    _step(m, t)
                    if reducer is not None:
                        reducer.finish()
                for _ in range(15):
                    one()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    one()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / steps

            plain = run(None)
            m.set_comm(red)
            with_native = run(red)
            q.put(("ok", plain, with_native, red.placement))
        red.close()
    except BaseException as e:
        q.put(("error", repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def _run_native(which):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_child, args=(which, q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert res[0] == "ok", res
    assert p.exitcode == 0
    return res


@pytest.mark.timeout(400)
def test_native_reducer_averages_every_block_inside_the_backward_call():
    """One rank over the library's own RCCL communicator (the collectives really run; mean over one rank = identity): with
    set_comm() the gradients after finish() equal the plain run's over several steps, accumulation into a live buffer is
    refused, and the stand-alone whole-buffer form returns the buffer unchanged."""
    _run_native("correct")


@pytest.mark.timeout(400)
def test_native_reducer_in_order_on_the_producing_streams():
    """dvgr_comm_set_inorder: the same gradients, each block's collective enqueued on the stream that wrote the block (the
    side stream for the layer blocks, the caller's for the query block); and the step costs no more than the other form."""
    _run_native("inorder_correct")
    _, plain, with_native, _ = _run_native("inorder_speed")
    print(f"  plain {plain * 1e3:.3f} ms, with the native reducer in order {with_native * 1e3:.3f} ms per step")
    assert with_native < 1.5 * plain, (plain, with_native)


@pytest.mark.timeout(400)
def test_native_reducer_costs_the_step_little():
    """Verdict of round 2: the data-parallel plumbing cost one rank 16 % (host callbacks into the interpreter, a helper
    thread beside the launch path, the query stream given up).  The native path keeps the query stream and its only host
    work is five enqueues inside the backward call (measured 1.05 x).  The failure this test guards against is the 3 x slow
    set-up of DESIGN.md section 5 (stream creation / RCCL set-up order), so the gate sits at 1.5 x; the measured overhead is
    printed and bench.py reports it."""
    _, plain, with_native, _ = _run_native("speed")
    print(f"  plain {plain * 1e3:.3f} ms, with the native reducer {with_native * 1e3:.3f} ms per step")
    assert with_native < 1.5 * plain, (plain, with_native)


@pytest.mark.timeout(400)
def test_a_communicator_created_late_is_placed_on_a_free_pipe():
    """Rounds 2-3 refused a communicator created after the unit had run (DVGR_E_ORDER: "3 x slower, not understood").  Round 4
    found the mechanism (two non-idle hardware queues on one command-processor pipe, tools/micro/pipe_probe.hip) and
    dvgr_comm_place picks the communication stream's queue by measurement: created LAST, behind the unit's streams and two
    more busy streams, the step with the reducer stays within the same gate as the early one."""
    if os.environ.get("DVGR_NO_STREAM_PLACEMENT", "0") == "1":
        pytest.skip("stream placement switched off in the environment")
    _, plain, with_native, placement = _run_native("late")
    print(f"  plain {plain * 1e3:.3f} ms, with the native reducer created late {with_native * 1e3:.3f} ms per step; {placement}")
    assert placement["candidates"] >= 2 and placement["measured_against"] >= 3
    assert placement["collisions_of_chosen"] < 2 ** (placement["measured_against"] - 1), placement    # the caller's stream is undisturbed
    assert with_native < 1.5 * plain, (plain, with_native)


@pytest.mark.timeout(600)
def test_two_ranks_graph_layers_2_reduce_the_whole_buffer_after_the_backward():
    """with graph_layers > 1 the backward refuses layer events (shared GAT modules: no block is
    final before the end) and layer_buckets() is ONE whole-buffer bucket that no callback ever launches -- TrainStep used to
    step on un-averaged unit gradients and the replicas diverged.  It now clears the per-block plumbing and reduces the
    whole flat buffer after the backward (behind the parameter buckets, one issue order): both ranks end with identical
    parameters and the averaged gradients equal a blocking all-reduce's."""
    import torch.multiprocessing as mp
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rendezvous = _rendezvous()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, rendezvous, q, 2)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, got, got_g, want_g, log = q.get(timeout=540)
        res[r] = (got, got_g, want_g, log)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][3] == res[1][3]                                             # the parameter buckets: one order
    assert all(t[0] == "param" for t in res[0][3])                            # (the unit's one collective is issued by reduce())
    assert np.array_equal(res[0][0], res[1][0])                               # replicas stay identical
    for r in range(world):
        scale = float(np.abs(res[r][2]).max())
        assert float(np.abs(res[r][1] - res[r][2]).max()) < 2e-5 * scale, r


@pytest.mark.timeout(900)
def test_bench_selftest_picks_its_reducer_from_child_trials():
    """Round-3 review: with BOTH reducers set up in one process the torch.distributed path measured 3.2 ms per step against
    1.2 alone, so an in-process trial could print a line that measures the set-up, not the path.  `bench.py` (reducer auto)
    now tries each candidate in a child process of its own before it touches the GPU and sets up the winner alone; a best
    candidate beyond 1.3 x the plain step is flagged `suspect_setup` (exit 3 under DVGR_BENCH_STRICT=1, which this self-test of
    the plumbing sets).  One rank, collectives really enqueued."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update({"DVGR_DDP_SELFTEST": "1", "MASTER_PORT": "29741", "MASTER_ADDR": "127.0.0.1", "DVGR_BENCH_STRICT": "1"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.5",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    dist_cfg = d["config"]["distributed"]
    trial = dist_cfg["reducer_trial"]["ms_per_step"]
    print("  trials", trial, "->", dist_cfg["reducer"][:40], "timed", d["ms_per_step"])
    assert trial["plain"] and (trial["native"] or trial["inorder"] or trial["python"])
    cands = {k: v for k, v in trial.items() if k != "plain" and v}
    best = min(cands, key=cands.get)
    assert dist_cfg["reducer"].startswith(best)
    assert cands[best] <= 1.3 * trial["plain"] and not dist_cfg["reducer_trial"].get("suspect_setup")
    assert d["ms_per_step"] <= 1.5 * trial["plain"]            # the timed region runs the winner alone: no 3 x artefact
    assert set(dist_cfg["allreduce_bytes_per_block"]) == {"0", "1", "2", "3", "4"}


@pytest.mark.timeout(900)
def test_bench_two_ranks_under_the_launcher_share_one_gpu():
    """The control flow of `bench.py --gpus N` for N > 1 as the driver starts it (python -m torch.distributed.run ... bench.py
    --gpus 2), rehearsed on ONE GPU: two ranks share the device (DVGR_BENCH_SHARE_GPU) over gloo.  Every rank runs its trial
    children (the native forms are skipped for gloo) on free ports shared over the launcher's agent store, the ranks agree on
    MAX per candidate after each one, the timed region runs the torch.distributed reducer and rank 0 prints ONE line with the
    whole-job value.  Gloo moves the 57 MB through the host, so the best candidate is far beyond 2.5 x the plain step: by
    default the line is printed WITH `suspect_setup` (a bandwidth-bound run is still a measurement); DVGR_BENCH_STRICT=1
    turns the same finding into exit code 3 and no line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["DVGR_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29761", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--min-seconds", "0.2", "--backend", "gloo", "--no-cpu-baseline"]
    env["DVGR_BENCH_STRICT"] = "1"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode != 0 and "suspect set-up" in r.stderr, r.stderr[-2000:]
    assert '"metric"' not in r.stdout
    del env["DVGR_BENCH_STRICT"]
    cmd[cmd.index("29761")] = "29771"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    dist_cfg = d["config"]["distributed"]
    print("  two ranks on one GPU over gloo:", d["ms_per_step"], "ms per step, trials", dist_cfg["reducer_trial"]["ms_per_step"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 64 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]       # whole-job QA-pairs/s
    assert dist_cfg["world_size"] == 2 and dist_cfg["reducer"].startswith("python")
    t = dist_cfg["reducer_trial"]["ms_per_step"]
    assert t["plain"] and t["python"] and t["native"] is None and t["inorder"] is None
    assert dist_cfg["reducer_trial"]["suspect_setup"] is True and dist_cfg["reducer_trial"]["best_over_plain"] > 2.5
