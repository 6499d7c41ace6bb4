"""GPU: the BASELINE.json configurations at their OWN dimensions (D = 768, word_dim = 300, T = 15), in exact fp32 and in
the benchmark's default operand mode (bf16x3), through the C ABI:

  C3  MSVD-QA    num_clips =  8, unit_layers =  3, batch 128
  C4  MSRVTT-QA  num_clips = 16, unit_layers =  6, batch 256 / 8 GPUs = 32 per GPU
  C5  stress     num_clips = 64, unit_layers = 12, batch 512 / 8 GPUs = 64 per GPU

C1 (batch 8: the 160-row small-batch schedule with 64-row tiles) and C2 (batch 64, the headline configuration) in eval AND
training mode as well -- training mode is what bench.py times; their golden-vector cases live in tests/test_hip_parity.py.
C3 and C4 are compared with the oracle at their full per-GPU batch,
eval and training mode (masks of the device RNG injected into the oracle).  C5's oracle comparison runs on a 2-sample
batch; its full batch of 64 is then tied to that comparison through properties the path offers: samples are independent
(rows of the full-batch outputs and input gradients equal the small-batch run's), the parameter gradient of a batch is
the sum of its shards' (linearity), and a run is deterministic.
Gates: 1e-4 forward, 5e-4 gradients (max |err| / max |want| per tensor), both operand modes."""
import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L
from dualvgr_videoqa_amd import dropout, synth
from helpers import aligned_oracle_backward, assert_flips_at_kink, assert_grads, device_logit_signs, rel_err
from oracle import unit_torch as O

pytestmark = pytest.mark.gpu
TOL, GTOL = 1e-4, 5e-4
D, WD, T = 768, 300, 15


def _module(N, Lr, dtype, sd_np, train):
    m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.cuda().train(train)
    m.keep_saved = True
    return m


def _upstream(o, R, N):
    """train.py:148-154 shaped scalar: a weighted read-out of vis + common + 1e-8 * HSIC over the list outputs."""
    c, d_ = O.aux_losses(o, N)
    return (o[0] * R).sum() + 0.3 * ((o[1] * R).sum() + (o[2] * R).sum()) + c + 1e-8 * d_


def _run(m, inp, R, N, seed=None):
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    if seed is not None:
        m.forced_seed = seed
    m.zero_grad(set_to_none=True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    _upstream(outs, R.cuda(), N).backward()
    torch.cuda.synchronize()
    return t, outs


def _against_oracle(name, B, N, Lr, dtype, train, scale_tol=None):
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=31, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=31)
    R = torch.from_numpy(synth.normal("Rcfg", (B, N, D), 1.0, 31))
    m = _module(N, Lr, dtype, sd_np, train)
    seed = 0x5EED_0000 + N if train else None
    t, outs = _run(m, inp, R, N, seed)
    masks = None
    if train:
        masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, B, N, D, Lr).items()}
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    signs = device_logit_signs(m, L.make_dims(B, N, T, D, WD, Lr), (B, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, masks, lambda o: _upstream(o, R, N), signs)
    flat_h = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    flat_o = [oo[0], oo[1], oo[2]] + oo[3] + oo[4] + oo[5] + oo[6]
    ferr = max(rel_err(h.detach().cpu(), o) for h, o in zip(flat_h, flat_o))
    what = f"{name} B={B} N={N} L={Lr} {dtype} {'train' if train else 'eval'}"
    print(f"  {what}: forward (all {len(flat_h)} outputs) {ferr:.2e}")
    assert ferr < TOL, ferr
    assert_flips_at_kink(stats, what)
    assert_grads(m, t, gin, gw, dbg, Lr, GTOL, what, scale_tol=scale_tol)
    return m, sd_np, t, outs


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_c1_cpu_reference_config_at_its_own_batch(dtype, train):
    """BASELINE.json configs[0]: B = 8 -> 160 rows at D = 768, the small-batch schedule (csrc/dvgr_api.hip: no query stream,
    64-row planes tiles, fp32 + planes layout because B N <= 256)."""
    _against_oracle("C1", 8, 20, 1, dtype, train)


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_c2_headline_config_full_batch(dtype, train):
    """BASELINE.json configs[1] at its full batch in the mode bench.py times (training, dropout masks of the device RNG
    injected into the oracle) and in eval mode."""
    _against_oracle("C2", 64, 20, 3, dtype, train)


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_c3_msvd_full_batch(dtype, train):
    _against_oracle("C3", 128, 8, 3, dtype, train)


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_c4_msrvtt_full_per_gpu_batch(dtype, train):
    _against_oracle("C4", 32, 16, 6, dtype, train)


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_c5_stress_oracle_on_a_shard_and_properties_at_full_batch(dtype):
    N, Lr, Bs, Bf = 64, 12, 2, 64
    # (1) the oracle comparison proper, on the first two samples of the full batch
    full = synth.unit_inputs(Bf, N, T, D, WD, seed=31)
    shard0 = {k: v[:Bs].copy() for k, v in full.items()}
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=31, bias_range=0.05)
    Rf = torch.from_numpy(synth.normal("Rc5", (Bf, N, D), 1.0, 31))

    def lin_upstream(o, R):          # linear in the outputs, so that gradients add over shards exactly
        return (o[0] * R).sum() + sum((x * R).sum() for x in o[3] + o[4] + o[5] + o[6]) + 0.3 * (o[1] * R).sum()

    m = _module(N, Lr, dtype, sd_np, False)
    t0 = {k: torch.from_numpy(v.copy()).cuda() for k, v in shard0.items()}
    for k in ("A", "M", "dq", "words"):
        t0[k].requires_grad_(True)
    outs0 = m(t0["A"], t0["M"], t0["dq"], t0["words"], t0["qlen"])
    lin_upstream(outs0, Rf[:Bs].cuda()).backward()
    torch.cuda.synchronize()
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in shard0.items()}
    signs = device_logit_signs(m, L.make_dims(Bs, N, T, D, WD, Lr), (Bs, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, None, lambda o: lin_upstream(o, Rf[:Bs]), signs)
    flat_h = [outs0[0], outs0[1], outs0[2]] + outs0[3] + outs0[4] + outs0[5] + outs0[6]
    flat_o = [oo[0], oo[1], oo[2]] + oo[3] + oo[4] + oo[5] + oo[6]
    ferr = max(rel_err(h.detach().cpu(), o) for h, o in zip(flat_h, flat_o))
    print(f"  C5 shard B={Bs} {dtype}: forward (all {len(flat_h)} outputs) {ferr:.2e}")
    assert ferr < TOL, ferr
    assert_flips_at_kink(stats, f"C5 shard {dtype}")
    # (bias gradients against their cancellation scale: 5e-5 is the 3-layer bound of bf16x3; through 12 layers, with the
    # projection on bf16x3 planes since round 4 -- the fused GAT-view kernel; the 128-row shard used to take the exact-fp32
    # batch-sized kernels there -- the worst one measures 6.3e-5)
    assert_grads(m, t0, gin, gw, dbg, Lr, GTOL, f"C5 shard {dtype}", scale_tol=None if dtype == "fp32" else 1.5e-4)
    g_shard0 = m.last_flat_grad.clone()
    out0_keep = [x.detach().clone() for x in (outs0[0], outs0[3][-1], outs0[6][-1])]
    gin0_keep = {k: t0[k].grad.clone() for k in ("A", "M", "dq", "words")}

    # (2) the full per-GPU batch of 64
    def run_full():
        t = {k: torch.from_numpy(v.copy()).cuda() for k, v in full.items()}
        for k in ("A", "M", "dq", "words"):
            t[k].requires_grad_(True)
        m.zero_grad(set_to_none=True)
        outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        lin_upstream(outs, Rf.cuda()).backward()
        torch.cuda.synchronize()
        return t, outs, m.last_flat_grad.clone()

    tf, outsf, gfull = run_full()
    # samples are independent: the first two rows of the full run are the shard run.  (A 128-row shard and the 4096-row
    # batch take different contraction kernels -- batch-sized exact-fp32 ones vs the ring kernels -- so the two runs agree
    # to the operand mode's rounding, not bit for bit.)
    stol = 2e-6 if dtype == "fp32" else 5e-5
    for a, b in zip(out0_keep, (outsf[0], outsf[3][-1], outsf[6][-1])):
        assert rel_err(b[:Bs].detach().cpu(), a.cpu()) < stol
    if dtype == "fp32":      # (gradients: only where both runs take the same LeakyReLU branches -- exact fp32 on both sides)
        for k in ("A", "M", "dq", "words"):
            assert rel_err(tf[k].grad[:Bs].cpu(), gin0_keep[k].cpu()) < 10 * stol, k
    # determinism of the forward; the backward's float atomics (split-K, column sums) may reorder
    tf2, outsf2, gfull2 = run_full()
    assert torch.equal(outsf[0], outsf2[0]) and torch.equal(outsf[5][-1], outsf2[5][-1])
    assert rel_err(gfull2.cpu(), gfull.cpu()) < 2e-6
    # linearity: the parameter gradient of the batch is the sum over its shards (8 shards of 8 samples)
    acc = torch.zeros_like(gfull)
    mag = torch.zeros_like(gfull)                # sum over shards of |shard gradient|: the scale the sums cancel from
    for s in range(0, Bf, 8):
        part = {k: v[s:s + 8].copy() for k, v in full.items()}
        tp = {k: torch.from_numpy(v).cuda() for k, v in part.items()}
        for k in ("A", "M", "dq", "words"):
            tp[k].requires_grad_(True)
        m.zero_grad(set_to_none=True)
        op = m(tp["A"], tp["M"], tp["dq"], tp["words"], tp["qlen"])
        lin_upstream(op, Rf[s:s + 8].cuda()).backward()
        acc += m.last_flat_grad
        mag += m.last_flat_grad.abs()
    torch.cuda.synchronize()
    # A consistency property (the parity gate is the oracle comparison above).  The two sides sum the same B N = 4096 row
    # terms in different orders, and the gradients of the deep layers cancel to ~1e-3 of their terms (rows of a deep GAT
    # stack share a large common component).  A cancellation scale would have to be the sum of the |row terms|, which only
    # the oracle knows (and only at the shard size); the device can offer the sum over the shards of |shard gradient|, and
    # a shard's own sum over its 512 rows cancels just as much -- measured against that scale the worst block reads 1.2e-2
    # in BOTH operand modes (profiles/experiments_r03.tsv era run, gpurun_out/r3b), i.e. it measures the cancellation, not
    # the arithmetic.  So the plain per-block metric stays, at bounds taken from the measured values with a 2 x margin:
    # exact fp32 9.3e-4 -> 2e-3 (as before), bf16x3 2.1e-3 -> 5e-3 (round 2 allowed 2e-2); the scale-relative figure is
    # printed beside it for the record.
    bound = 2e-3 if dtype == "fp32" else 5e-3
    worst_all = []
    for l in range(Lr + 2):                      # per block of the flat buffer
        a, b = m.layer_param_range(l)
        scale = mag[a:b].clamp_min(5e-2 * float(mag[a:b].max()))
        rel_scale = float(((acc[a:b] - gfull[a:b]).abs() / scale).max())
        plain = rel_err(acc[a:b].cpu(), gfull[a:b].cpu())
        print(f"  C5 {dtype} linearity block {l}: plain metric {plain:.2e} (|err| / sum of |shard gradients| {rel_scale:.2e})")
        worst_all.append((plain, l, rel_scale))
    assert max(worst_all)[0] < bound, max(worst_all)
    assert torch.isfinite(gfull).all() and float(gfull.abs().max()) > 0
    del g_shard0


def test_c5_training_mode_shard_on_the_kernels_the_full_batch_runs():
    """Round-3 review: C5's oracle comparison ran in eval mode on a 2-sample shard (128 rows: the batch-sized exact-fp32
    contractions), while bench.py --config c5 times TRAINING mode on the ring / wide / 8-wave N = 64 kernels.  Here: N = 64,
    L = 12, D = 768 (F = 192 per head), bf16x3, training mode, on a 5-sample shard = 320 rows -- above the batch-sized
    kernels' range and inside the plane-only layout (csrc/dvgr_api.hip: lean_planes), so the contractions, the fused GAT-view
    kernel (8 row tiles x 2 samples), the attention backward <4, 8> and the staged loss Gram are the ones the full batch
    takes -- against the oracle with the device's 24 x 12 dropout masks injected, forward and every gradient.  Then the
    full per-GPU batch of 64 in training mode with the same seed: the dropout counters of a sample's elements do not depend
    on the batch size, so the first five samples of the full run must reproduce the shard run."""
    N, Lr, Bs, Bf, dtype = 64, 12, 5, 64, "bf16x3"
    m, sd_np, t0, outs0 = _against_oracle("C5-shard", Bs, N, Lr, dtype, True, scale_tol=1.5e-4)   # (12 layers deep: see the eval-mode shard)
    seed = 0x5EED_0000 + N
    keep = [x.detach().clone() for x in (outs0[0], outs0[3][-1], outs0[6][-1], outs0[4][0])]
    full = synth.unit_inputs(Bf, N, T, D, WD, seed=31)
    shard = synth.unit_inputs(Bs, N, T, D, WD, seed=31)
    if not all(np.array_equal(full[k][:Bs], shard[k]) for k in full):
        pytest.skip("synthetic inputs of the shard are not a prefix of the full batch's")
    tf = {k: torch.from_numpy(v.copy()).cuda() for k, v in full.items()}
    m.forced_seed = seed
    with torch.no_grad():
        outsf = m(tf["A"], tf["M"], tf["dq"], tf["words"], tf["qlen"])
    torch.cuda.synchronize()
    for a, b in zip(keep, (outsf[0], outsf[3][-1], outsf[6][-1], outsf[4][0])):
        e = rel_err(b[:Bs].detach().cpu(), a.cpu())
        assert e < 5e-5, e
    assert float((outsf[3][0] == 0).float().mean()) > 0.1          # dropout ran


def test_c2_xcd_block_map_changes_no_result():
    """Round 6: the contraction engine's XCD block map (GemmBatch::xmap; dvgr_dbg_xmap_tags) against the default workgroup -> tile
    map on the headline shape, training mode: the forward is bit-identical, every gradient equal up to the order of the float
    atomics that exist in either form (column sums, loss scalars)."""
    cfg = dict(B=64, N=20, L=3, T=15)
    B, N, Lr, T = cfg["B"], cfg["N"], cfg["L"], cfg["T"]
    sd_np = synth.xavier_state_dict(O.param_shapes(300, 768, Lr), seed=61, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, 768, 300, seed=61)
    lib = L.load()
    got = {}
    prev = lib.dvgr_dbg_xmap_tags(0)
    try:
        for mask in (0, 127):
            lib.dvgr_dbg_xmap_tags(mask)
            m = dv.DualVGRUnit_multiple(word_dim=300, module_dim=768, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
            m = m.cuda().train()
            m.forced_seed = 977
            t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
            for k in ("A", "M", "dq", "words"):
                t[k].requires_grad_(True)
            outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
            O.bench_scalar_loss(outs, N).backward()
            torch.cuda.synchronize()
            got[mask] = ([outs[0].detach().cpu(), outs[3][-1].detach().cpu(), outs[6][-1].detach().cpu()],
                         {k: t[k].grad.detach().cpu() for k in ("A", "M", "dq", "words")},
                         {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    finally:
        lib.dvgr_dbg_xmap_tags(prev if prev >= 0 else -1)
    for a, b in zip(got[0][0], got[127][0]):
        assert torch.equal(a, b)
    # (float atomics whose order differs from run to run -- the batched d(dq), the column sums -- feed the query block: two runs of
    # ONE map differ by ~2e-6 there as well)
    for k in got[0][1]:
        assert rel_err(got[127][1][k], got[0][1][k]) < 1e-5, k
    for k in got[0][2]:
        if k.endswith(".bias"):     # column sums by float atomics (cancelling: two runs of ONE map differ by 2e-5 there); a.bias / fc.bias are zero in exact arithmetic
            assert rel_err(got[127][2][k], got[0][2][k]) < 5e-4 or k.endswith((".fc.bias", ".a.bias")), k
            continue
        assert rel_err(got[127][2][k], got[0][2][k]) < 1e-5, k
