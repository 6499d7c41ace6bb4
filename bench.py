#!/usr/bin/env python3
"""Benchmark of the DualVGR reasoning unit on MI355X: QA-pairs/sec (forward+backward) per GPU,
SVQA num_clips=20 (BASELINE.json).  One step = one forward + one backward of the stacked unit
over one batch of synthetic inputs already resident in HBM (training mode, dropout on).

    python bench.py [--gpus N] [--steps K] [--warmup W]           # N > 1 via torch.distributed.run

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job QA-pairs/s, the
roofline of the dominant kernel (forward GAT projection GEMM, timed live with HIP events on the
launch stream) and the CPU baseline (the oracle's op-for-op PyTorch-CPU restatement of the
reference, timed on this box's host cores, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

# BASELINE.json `configs` (per-GPU batch; D = 768, Wd = 300, T = 15 as in SURVEY.md section 8)
CONFIGS = {
    "c1": dict(B=8, N=20, L=1, T=15, name="SVQA num_clips=20 unit_layers=1 batch=8"),
    "c2": dict(B=64, N=20, L=3, T=15, name="SVQA num_clips=20 unit_layers=3 batch=64"),
    "c3": dict(B=128, N=8, L=3, T=15, name="MSVD-QA num_clips=8 unit_layers=3 batch=128"),
    "c4": dict(B=32, N=16, L=6, T=15, name="MSRVTT-QA num_clips=16 unit_layers=6 batch=256/8 per GPU"),
    "c5": dict(B=64, N=64, L=12, T=15, name="stress num_clips=64 unit_layers=12 batch=512/8 per GPU"),
}
D, WD = 768, 300
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}     # /opt/skills/guides/MI355X_MICROARCH.md, dense MFMA


def f_fwd(N, L, T):
    """Algorithmic GEMM FLOPs per QA pair, forward."""
    return L * (16 * N * D * D + 2 * T * D * D + 8 * N * N * D + 28 * N * D + 2 * T * D + 2 * T * WD + 4 * WD * D) + 2560 * N * D


def cpu_baseline(cfg, seconds=15.0):
    """Reference CPU path (oracle = op-for-op port, kind 'port') on the host cores: forward+backward
    of the unit at the benchmark's N / L / T on a bounded sample of the batch."""
    from oracle import unit_torch as O          # test infrastructure, used only as the timed baseline
    from dualvgr_videoqa_amd import synth
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    Bs = 8
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in
          synth.xavier_state_dict(O.param_shapes(WD, D, cfg["L"]), seed=666).items()}
    inp = synth.unit_inputs(Bs, cfg["N"], cfg["T"], D, WD, seed=666)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    masks = None

    def step():
        for p in sd.values():
            p.grad = None
        outs = O.unit_forward(sd, t["A"], t["M"], t["dq"], t["words"], t["qlen"], cfg["L"], 1, masks=masks, naive=True)
        O.bench_scalar_loss(outs, cfg["N"]).backward()

    # PyTorch-CPU does not scale these small ops over a large host: give the reference its best
    # thread count among a few candidates (one warm-up + one timed step each), then time that one
    best = None
    for th in [c for c in (8, 16, 32, 64) if c <= cores] or [cores]:
        torch.set_num_threads(th)
        step()
        t0 = time.perf_counter()
        step()
        one = time.perf_counter() - t0
        print(f"[cpu_baseline] {th} threads: {one * 1e3:.0f} ms/step", file=sys.stderr, flush=True)
        if best is None or one < best[1]:
            best = (th, one)
        if one > 20.0:
            break
    cores, one = best
    torch.set_num_threads(cores)
    n = int(max(3, min(200, seconds / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {"value": Bs / dt, "unit": "QA-pairs/s", "cores": cores, "kind": "port",
            "sample": f"B={Bs} of the {cfg['B']}-pair batch, N={cfg['N']} L={cfg['L']} T={cfg['T']} fp32 eval-mode "
                      f"forward+backward, naive (B,N,N,2F) form, 2 warm-up + {n} timed steps, {dt * 1e3:.1f} ms/step, "
                      f"{cores} threads on {model}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="MFMA operand precision: bf16x3 (default; fp32 split hi+lo, meets the 1e-4 fp32 parity gate), "
                         "fp32 (exact fp32 MFMA), bf16 (plain bf16 operands, 3e-2 tolerance)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (default: training mode, dropout on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--probe", default="fwd_gat_gemm")
    ap.add_argument("--head", action="store_true",
                    help="row f2: also run the clip read-out + classifier (AnswerHead, 1000 answers) and a cross-entropy "
                         "loss on its logits inside every step; the unit then gets d CE / d vis instead of ones")
    ap.add_argument("--model", action="store_true",
                    help="row f3: the whole DualVGR model -- appearance BiLSTM, unit and answer head on HIP, question BiLSTMs "
                         "and the motion Linear on stock PyTorch-ROCm (MIOpen, hipBLASLt) -- on raw (B, N, 16, 2048) / "
                         "(B, N, 2048) features, cross-entropy + auxiliary losses, forward + backward")
    ap.add_argument("--question-encoder", default="hip", choices=["hip", "miopen"],
                    help="with --model: the two packed question BiLSTMs on this package's HIP path (default) or on nn.LSTM / MIOpen")
    ap.add_argument("--appearance-encoder", default="hip", choices=["hip", "miopen"],
                    help="with --model: the appearance BiLSTM on this package's HIP path (default) or on nn.LSTM / MIOpen")
    ap.add_argument("--loss", default="fused", choices=["fused", "fused-after", "precomputed"],
                    help="fused: the scope-U loss and its gradients are computed inside every timed step by "
                         "dvgr_scope_u_loss; precomputed: fixed upstream gradients (unit kernels only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU fallback of the product path)")
    local = min(local, torch.cuda.device_count() - 1)          # (testing aid: several ranks may share one GPU with gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # DVGR_DDP_SELFTEST=1: run the data-parallel plumbing (process group, per-layer buckets gated by the library's events,
    # communication stream) with a single rank, to check it on a one-GPU box
    ddp = world > 1 or os.environ.get("DVGR_DDP_SELFTEST", "0") == "1"
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)   # RCCL's own stream on a queue of its own as well
            dist.init_process_group("nccl", device_id=dev, pg_options=opts)
        else:
            dist.init_process_group(args.backend)

    import dualvgr_videoqa_amd as dv
    from dualvgr_videoqa_amd import _lib as L_
    from dualvgr_videoqa_amd import losses, parallel, synth

    cfg = CONFIGS[args.config]
    B, N, Lr, T = cfg["B"], cfg["N"], cfg["L"], cfg["T"]
    if args.model:
        bench_model(args, dev, rank, world, cfg)
        if ddp:
            dist.destroy_process_group()
        return
    shapes = {n: s for n, s, _ in dv.param_table(L_.make_dims(1, N, 1, D, WD, Lr))}
    sd_np = synth.xavier_state_dict(shapes, seed=666)          # same weights on every rank (as after a broadcast)
    m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=args.dtype)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m = m.to(dev)
    m.train(not args.eval_mode)
    if args.loss == "fused":
        m.loss_in_forward = (1.0, 1e-8, float(world), not args.head)
    head = None
    if args.head:
        if args.loss != "fused":
            raise SystemExit("--head needs --loss fused")
        torch.manual_seed(666)
        head = dv.AnswerHead(module_dim=D, num_answers=1000, dtype=args.dtype).to(dev)
        head.train(not args.eval_mode)
        qe = torch.from_numpy(synth.normal("bench_qe", (B, D), 0.5, 666 + rank)).to(dev).requires_grad_(True)
        answers = torch.from_numpy((synth.uniform("bench_ans", (B,), 0.0, 1000.0, 666 + rank)).astype(np.int64)).to(dev)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=666 + rank)  # each rank owns its own QA pairs (weak scaling)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)                              # the encoders upstream need these gradients

    # upstream gradients of every output: the true gradient of the scope-U loss
    # sum(vis) + common + 1e-8 * HSIC at the first forward, held fixed for the timed
    # steps (the auxiliary-loss arithmetic is caller code, not the path: SURVEY.md 8f row f1)
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    det = [r.detach().clone().requires_grad_(True) for r in raw]
    loss0 = losses.scope_u_loss((det[0], det[1], det[2], list(det[3].unbind(0)), list(det[4].unbind(0)),
                                 list(det[5].unbind(0)), list(det[6].unbind(0))), N, world_size=world)
    ups = list(torch.autograd.grad(loss0, det, allow_unused=True))
    ups = [u if u is not None else torch.zeros_like(r) for u, r in zip(ups, raw)]
    del raw, det, loss0

    events = None
    reducer = None
    if ddp:
        events = [torch.cuda.Event() for _ in range(Lr + 1)]
        m.set_layer_events(events)
        reducer = parallel.LayerBucketReducer(parallel.layer_buckets(m), device=dev)
        reducer.force = world == 1          # self-test: run the collectives even with one rank
        # each bucket's all-reduce is enqueued from inside the backward call, right after its event (dvgr_opts.layer_done_cb)
        m.set_layer_callback(lambda idx, g: reducer.launch(idx, g, events[idx]))

    probe_sites = {"fwd_gat_gemm": (L_.PROBE_FWD_GAT_GEMM, 8 * B * N * D * D, "gemm_planes_kernel<KC,KC,TAG_FWD_GAT=1,TERMS> / gemm_ring_kernel<KC,KC,1,F32> (Wh = X.Wcat^T, both modalities)"),
                   "fwd_sf_gemm": (L_.PROBE_FWD_SF_GEMM, 8 * B * N * D * D, "gemm_kernel<KC,KC,*,*,TAG_FWD_SF=2>"),
                   "bwd_gat_dw": (L_.PROBE_BWD_GAT_DW, 8 * B * N * D * D, "gemm_kernel<IC,IC,*,*,TAG_BWD_GAT_DW=3>"),
                   "bwd_gat_dx": (L_.PROBE_BWD_GAT_DX, 8 * B * N * D * D, "gemm_kernel<KC,IC,*,*,TAG_BWD_GAT_DX=4>")}
    site, probe_flops, probe_name = probe_sites[args.probe]
    # layer 0's Wh contraction shares the chip with the query-side chain on the side stream; layer 1's runs alone
    probe_layer = min(1, Lr - 1)
    # timing events without the system-scope fence at their record (their documented use): closer to the kernel's own duration
    pairs = [(L_.ProbeEvent(), L_.ProbeEvent()) for _ in range(args.steps)]
    for ev in (events or []):
        ev.record()          # torch creates the hipEvent lazily at the first record; the C ABI needs the handle

    def step(i=None):
        if i is not None:
            m.set_probe(site, probe_layer, pairs[i][0], pairs[i][1])
        m.zero_grad_flat()
        if reducer is not None:
            reducer.start(Lr)
        for k in ("A", "M", "dq", "words"):
            t[k].grad = None
        raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        if args.loss == "fused":
            # loss value + its gradients w.r.t. all outputs by the fused kernels (csrc/dvgr_loss.hip), every step;
            # enqueued by the forward itself (dvgr_opts.loss) so that they run under its last contractions
            _, g = m.last_loss
            if head is not None:
                head.zero_grad(set_to_none=True)
                qe.grad = None
                logits = head(qe, raw[0])
                _, g_logits = head.cross_entropy(logits, answers)                     # train.py:147, fused value + gradient
                torch.autograd.backward([logits, *raw[3:]], [g_logits, *g[3:]])
            else:
                torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])
        elif args.loss == "fused-after":
            _, g = losses.fused_scope_u_loss(raw, 1.0, 1e-8, world)
            torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])
        else:
            torch.autograd.backward(raw, ups)
        if reducer is not None:
            reducer.finish()

    for _ in range(args.warmup):
        step()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    m.set_probe(0)

    ms_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    probe_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))
    achieved = probe_flops / (probe_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    f_train = 3 * f_fwd(N, Lr, T)
    # HBM bytes per launch of the probed kernel from PMC (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate
    # passes; profiles/r01_v8_bf16x3_c2_pmc_hbm.txt): measured for this one configuration only
    traffic = 90.0e6 if (args.config == "c2" and args.dtype == "bf16x3" and args.probe == "fwd_gat_gemm"
                         and not args.eval_mode) else None
    out = {
        "metric": "QA-pairs/sec (forward+backward) per GPU, SVQA num_clips=20",
        "value": value, "unit": "QA-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3 (fp32 split hi+lo, 3 bf16 MFMAs, fp32 accumulate)"}[args.dtype], "data": "synthetic",
        "config": {"workload": f"{cfg['name']}: DualVGR unit forward+backward, per-GPU batch {B}, D={D}, word_dim={WD}, "
                               f"T={T}, {'eval' if args.eval_mode else 'training (dropout 0.15 on)'}",
                   "global_batch": world * B, "parallelism": f"dp{world}",
                   "head": "AnswerHead (read-out + classifier, 1000 answers) + cross-entropy inside the step" if args.head else None,
                   "loss": ("sum(vis)+common+1e-8*HSIC evaluated with its gradients inside every timed step "
                            "(fused HIP kernels, dvgr_opts.loss / dvgr_scope_u_loss)") if args.loss.startswith("fused") else
                           "upstream gradients of sum(vis)+common+1e-8*HSIC for all 3+4L outputs, precomputed once",
                   "algorithmic_gflop_per_qa_train": f_train / 1e9,
                   "unit_tflops": value / world * f_train / 1e12,
                   "unit_frac_of_mfma_peak": value / world * f_train / 1e12 / peak},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "kernel": probe_name, "launch_ms": probe_ms,
                     "algorithmic_gflop_per_launch": probe_flops / 1e9,
                     # SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD of this kernel, collected with
                     # rocprofv3 --pmc at C2; not re-measured by this run
                     "mfma_pipe_busy_pmc": ({"bf16x3": 0.20, "fp32": 0.55}.get(args.dtype)
                                            if args.config == "c2" and args.probe == "fwd_gat_gemm" else None)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if ddp:
        dist.destroy_process_group()


def bench_model(args, dev, rank, world, cfg):
    """Scope M: the whole model at the DualVGR.forward boundary (synthetic inputs as listed there)."""
    import dualvgr_videoqa_amd as dv
    from dualvgr_videoqa_amd import synth
    B, N, Lr, T = cfg["B"], cfg["N"], cfg["L"], cfg["T"]
    V, FR, VQ, VA = 2048, 16, 4000, 1000
    vocab = {"question_token_to_idx": {str(i): i for i in range(VQ)}, "answer_token_to_idx": {str(i): i for i in range(VA)}}
    torch.manual_seed(666)
    m = dv.DualVGR(vision_dim=V, module_dim=D, word_dim=WD, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1,
                   unit_layers=Lr, dtype=args.dtype).to(dev)
    if "miopen" in (args.appearance_encoder, args.question_encoder):      # A/B timing against MIOpen's nn.LSTM (tools/)
        from tools.stock_encoders import use_stock_encoders
        use_stock_encoders(m, appearance=args.appearance_encoder == "miopen", question=args.question_encoder == "miopen")
    m.train(not args.eval_mode)
    m.overlap_encoders = os.environ.get("DVGR_NO_ENCODER_OVERLAP", "0") != "1"
    unit, head = m.visual_input_unit, m._head
    unit.loss_in_forward = (1.0, 1e-8, float(world), False)
    app = torch.from_numpy(np.abs(synth.normal("b_app", (B, N, FR, V), 0.5, 666 + rank))).to(dev)
    mot = torch.from_numpy(np.abs(synth.normal("b_mot", (B, N, V), 0.5, 666 + rank))).to(dev)
    qlen = torch.from_numpy((3 + (synth.uniform("b_ql", (B,), 0.0, 1.0, 666 + rank) * (T - 2)).astype(np.int64)).clip(1, T)).to(dev)
    qlen[0] = T
    q = torch.from_numpy((2 + (synth.uniform("b_q", (B, T), 0.0, 1.0, 666 + rank) * (VQ - 2)).astype(np.int64)).clip(2, VQ - 1)).to(dev)
    q = q * (torch.arange(T, device=dev)[None, :] < qlen[:, None])
    answers = torch.from_numpy((synth.uniform("b_ans", (B,), 0.0, float(VA), 666 + rank)).astype(np.int64)).to(dev)

    def step():
        m.zero_grad(set_to_none=True)
        logits, _, _, ca, cm, aq, mq = m(app, mot, q, qlen)
        _, g_logits = head.cross_entropy(logits, answers, scale=1.0 / world)
        _, g = unit.last_loss
        torch.autograd.backward([logits, *ca, *cm, *aq, *mq],
                                [g_logits, *g[3].unbind(0), *g[4].unbind(0), *g[5].unbind(0), *g[6].unbind(0)])
        if world > 1:      # data parallel: one flattened all-reduce of every gradient (losses are already scaled by 1/world)
            grads = [p.grad for p in m.parameters() if p.grad is not None]
            flat = torch._utils._flatten_dense_tensors(grads)
            dist.all_reduce(flat)
            for gr, red in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
                gr.copy_(red)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"metric": "QA-pairs/sec (forward+backward) per GPU, whole model (scope M), SVQA num_clips=20",
           "value": world * B * args.steps / dt, "unit": "QA-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"{cfg['name']}: whole DualVGR model forward+backward, per-GPU batch {B}, raw features "
                                  f"(B,{N},{FR},{V}) + (B,{N},{V}), T={T}, vocab {VQ}/{VA}; appearance encoder on {args.appearance_encoder}, question encoder on {args.question_encoder}, unit + head on hip"}}
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
