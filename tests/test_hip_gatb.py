"""GPU: the fused GAT-view BACKWARD kernel (csrc/dvgr_gatb.hip: the S4 data contraction dz_v += dpre_v . P1 with the attention
backward of model/GraphNN.py:95-111 as its epilogue, one launch per layer; model/Attention.py:20-23 backward) against

  (1) the two-launch form of the same stage (DVGR_OPT_NO_FUSED_GAT_BWD: gemm_planes_kernel<TAG_BWD_SF_DX>, then
      k_gat_attn_bwd) -- every input and parameter gradient to fp32 rounding (same operand planes, same MFMA chain per
      element of the contraction; the attention's reductions run in a different order);
  (2) the oracle, every gradient, at tile shapes the other test files do not reach: a last tile with fewer samples than the
      others, N not a multiple of 4, both row-tile instantiations (4 and 5 row tiles), single-stream mode, eval and training
      mode (all dropout sites: the alpha-site and out-site masks are recomputed inside the epilogue).
Shapes the kernel does not take (N > 20 at D = 768: the per-sample images of whole samples do not fit one compute unit's LDS;
graph_layers > 1) must run the two-launch form unchanged.  All through the C ABI via the drop-in module, D = 768, bf16x3
(the bench's mode) and plain bf16."""
import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L
from dualvgr_videoqa_amd import dropout, synth
from helpers import aligned_oracle_backward, assert_flips_at_kink, assert_grads, device_logit_signs, rel_err
from oracle import unit_torch as O

pytestmark = pytest.mark.gpu
D, WD, T = 768, 300, 9
TOL, GTOL = 1e-4, 5e-4


def _module(N, Lr, dtype, sd_np, train, fused_bwd, single=False, G=1, reads_dz=False, lo8=False, kb64=True):
    m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=dtype,
                                appearance_graph_layers=G, motion_graph_layers=G)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.cuda().train(train)
    m.keep_saved = True
    m.fused_gat_bwd = "force" if fused_bwd else False      # every supported shape, not only the ones it is faster at
    m.single_stream = single
    m.lo8_planes = lo8                    # True: 8-bit lo planes of the weight operands in the fused kernels (dvgr_opts.lo8; 32-deep tiles)
    m.kb64 = kb64                         # False: 32-deep K-tiles (DVGR_OPT_KB32)
    m.fused_bwd_reads_dz = reads_dz       # True: k_sf_bwd stores dz's pre-part and the fused launch reads it (DVGR_OPT_FUSED_BWD_READS_DZ)
    return m


def _scalar(o, Rr, N):
    c, d_ = O.aux_losses(o, N)
    return (o[0] * Rr).sum() + 0.3 * ((o[1] * Rr).sum() + (o[2] * Rr).sum()) + c + 1e-8 * d_


def _grads(m, inp, R, N, seed):
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    m.forced_seed = seed
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    _scalar(outs, R.cuda(), N).backward()
    torch.cuda.synchronize()
    g = {k: t[k].grad.detach().cpu().clone() for k in ("A", "M", "dq", "words")}
    g.update({"p:" + k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    return outs, t, g


# (B, N, L): N = 20 -> 5 row tiles x 4 samples, B = 7 leaves a last tile of 3 samples; N = 8 / 16 / 13 -> 4 row tiles (13 is not a
# multiple of 4: reduction steps past the sample's rows); B = 1 < samples per tile; B = 33, N = 20: 660 rows, lean planes
# (dWh / dpre exist as planes only, as at the bench's batch)
# N = 17 / 18 / 19: 5 row tiles whose epilogue image is LARGER than the K ring, so the launch's LDS size comes from the epilogue
# layout alone (GatbEpi::floats -- six per-row arrays; round 6: the host once counted five and only the allocation granule hid it)
SHAPES = [(7, 20, 2), (17, 8, 2), (9, 16, 2), (10, 13, 2), (1, 20, 2), (33, 20, 1), (40, 8, 1), (9, 17, 2), (8, 18, 2), (7, 19, 2)]


@pytest.mark.parametrize("B,N,Lr", SHAPES)
@pytest.mark.parametrize("train", [False, True])
def test_fused_backward_equals_two_launch_form(B, N, Lr, train):
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=51, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=51)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 51))
    lib = L.load()
    got, launches = {}, {}
    for fused in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, train, fused)
        n0 = lib.dvgr_dbg_launch_count()
        _, _, got[fused] = _grads(m, inp, R, N, 778)
        launches[fused] = lib.dvgr_dbg_launch_count() - n0
    worst, worst_k = 0.0, None
    for k in got[True]:
        if k.endswith((".fc.bias", ".a.bias")):   # gradients that are zero in exact arithmetic (softmax shift invariance): rounding noise
            continue
        e = rel_err(got[True][k], got[False][k])
        if e > worst:
            worst, worst_k = e, k
    print(f"  fused backward vs two launches, B={B} N={N} train={train}: worst rel err {worst:.2e} ({worst_k}); "
          f"launches {launches[True]} against {launches[False]}")
    assert worst < 5e-5, (worst, worst_k)
    # one launch per layer fewer, one more for the transposed P1 planes
    assert launches[False] - launches[True] == Lr - 1, launches


@pytest.mark.parametrize("B,N,Lr,single,train", [(6, 20, 2, False, True), (6, 20, 2, True, True), (5, 13, 2, True, True), (9, 8, 3, False, True),
                                                (7, 16, 2, False, False), (70, 20, 1, False, True), (5, 18, 2, False, True), (6, 19, 2, False, True)])
def test_fused_backward_against_oracle(B, N, Lr, single, train):
    """Every gradient against the oracle with the device's dropout masks injected (the kink-aware gate of tests/helpers.py)."""
    dtype = "bf16x3"
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=53, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=53)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 53))
    m = _module(N, Lr, dtype, sd_np, train, True, single=single)
    seed = 4343
    outs, t, _ = _grads(m, inp, R, N, seed)
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, B, N, D, Lr).items()} if train else None
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    signs = device_logit_signs(m, L.make_dims(B, N, T, D, WD, Lr), (B, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, masks, lambda o: _scalar(o, R, N), signs)
    what = f"gatb B={B} N={N} L={Lr} {dtype} single={single} train={train}"
    assert_flips_at_kink(stats, what)
    assert_grads(m, t, gin, gw, dbg, Lr, GTOL, what)


@pytest.mark.parametrize("B,N,Lr,train", [(7, 20, 2, True), (17, 8, 2, False), (10, 13, 3, True)])
def test_both_sources_of_dz_agree(B, N, Lr, train):
    """The fused launch forms dz's pre-part itself (default) or reads what k_sf_bwd stored (DVGR_OPT_FUSED_BWD_READS_DZ): the same
    arithmetic, beta_v (dX' + g_embed) + the list gradient, evaluated in two places; L = 3 covers a middle layer (no g_embed)."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=57, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=57)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 57))
    got = {}
    for reads in (False, True):
        _, _, got[reads] = _grads(_module(N, Lr, "bf16x3", sd_np, train, True, reads_dz=reads), inp, R, N, 99)
    worst = max(rel_err(got[True][k], got[False][k]) for k in got[True] if not k.endswith((".fc.bias", ".a.bias")))
    assert worst < 5e-5, worst


@pytest.mark.parametrize("kb64,lo8", [(False, False), (False, True)])
def test_fused_backward_32_deep_forms_against_oracle(kb64, lo8):
    """The 32-deep K-tile forms of the fused kernels (DVGR_OPT_KB32), with the bf16 and with the 8-bit lo plane of the weight
    operands, hold the same gates against the oracle as the default 64-deep form."""
    B, N, Lr = 6, 20, 2
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=59, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=59)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 59))
    m = _module(N, Lr, "bf16x3", sd_np, True, True, single=True, kb64=kb64, lo8=lo8)
    seed = 4545
    outs, t, _ = _grads(m, inp, R, N, seed)
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, B, N, D, Lr).items()}
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    signs = device_logit_signs(m, L.make_dims(B, N, T, D, WD, Lr), (B, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, masks, lambda o: _scalar(o, R, N), signs)
    what = f"gatb bf16x3 kb64={kb64} lo8={lo8}"
    flat_h = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    flat_o = [oo[0], oo[1], oo[2]] + oo[3] + oo[4] + oo[5] + oo[6]
    ferr = max(rel_err(h.detach().cpu(), o) for h, o in zip(flat_h, flat_o))
    print(f"  {what}: forward {ferr:.2e}")
    assert ferr < TOL, ferr
    assert_flips_at_kink(stats, what)
    assert_grads(m, t, gin, gw, dbg, Lr, GTOL, what)


def test_shapes_the_kernel_does_not_take_keep_two_launches():
    """N = 24 (6 row tiles: the images of 4 whole samples exceed 160 KB) and graph_layers = 2: the flag changes nothing."""
    lib = L.load()
    for (B, N, Lr, G) in [(6, 24, 2, 1), (6, 16, 2, 2)]:
        sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr, G), seed=54, bias_range=0.05)
        inp = synth.unit_inputs(B, N, T, D, WD, seed=54)
        R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 54))
        got, launches = {}, {}
        for fused in (True, False):
            m = _module(N, Lr, "bf16x3", sd_np, True, fused, G=G)
            n0 = lib.dvgr_dbg_launch_count()
            _, _, got[fused] = _grads(m, inp, R, N, 5)
            launches[fused] = lib.dvgr_dbg_launch_count() - n0
        assert launches[True] == launches[False], (N, G, launches)
        for k in got[True]:           # (not bit-equal run to run: d(dq) and the small-batch weight gradients add partial tiles atomically)
            if not k.endswith((".fc.bias", ".a.bias")):
                assert rel_err(got[True][k], got[False][k]) < 5e-5, (N, G, k)


def test_plain_bf16_mode_takes_the_fused_backward_too():
    B, N, Lr = 7, 20, 2
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=55, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=55)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 55))
    got = {}
    for fused in (True, False):
        _, _, got[fused] = _grads(_module(N, Lr, "bf16", sd_np, True, fused), inp, R, N, 6)
    # one plane per operand: an fp32-rounding difference upstream can flip the bf16 rounding of a later operand element
    worst = max(rel_err(got[True][k], got[False][k]) for k in got[True] if not k.endswith((".fc.bias", ".a.bias")))
    assert worst < 5e-3, worst


def test_race_screen_of_the_64_deep_rings_at_the_bench_size():
    """The 2-slot rings of 64-deep K-tiles (fused forward, fused backward, the engine's NT / NN launches) order LDS-DMA writes
    and fragment reads by counted vmcnt + one barrier per K-tile; a read that beat its DMA would show as a rare wrong tile, not
    as a failing reference check.  So: C2's own size (B = 64, N = 20, 3 layers: 256 workgroups, every CU busy), training mode,
    60 repetitions of forward + backward with one seed while a second stream keeps the fabric busy -- the forward outputs and
    the saved projections must be BIT-identical to the first repetition's every time (no atomics on that path), every gradient
    equal to 2e-4 (column sums and the punish gate's gradient are float atomic adds: measured 6e-5 between repetitions)."""
    B, N, Lr = 64, 20, 3
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=61, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=61)
    R = torch.from_numpy(synth.normal("R_race", (B, N, D), 1.0, 61))
    dims = L.make_dims(B, N, T, D, WD, Lr)
    m = _module(N, Lr, "bf16x3", sd_np, True, True)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    first, worst, worst_k = None, 0.0, None
    for it in range(60):
        with torch.cuda.stream(side):                      # ~ 128 MB of copies per repetition beside the step
            noise[: 32 << 20].copy_(noise[32 << 20:])
        m.zero_grad(set_to_none=True)
        outs, _, g = _grads(m, inp, R, N, 779)
        fw = [outs[0].cpu(), outs[1].cpu(), outs[2].cpu()] + [x.cpu() for lst in outs[3:] for x in lst]
        off = L.saved_offset(dims, L.S_WH, Lr - 1, 0)
        fw.append(m.last_saved[off:off + 4 * B * N * 8 * (D // 4)].cpu().clone())
        if first is None:
            first = (fw, g)
            continue
        for a, b in zip(fw, first[0]):
            assert torch.equal(a, b), f"repetition {it}: a forward tensor differs from the first repetition's"
        for k in g:
            if k.endswith((".fc.bias", ".a.bias")):
                continue
            e = rel_err(g[k], first[1][k])
            if e > worst:
                worst, worst_k = e, k
        assert worst < 2e-4, (it, worst, worst_k)         # (a wrong 80 x 192 tile would be an error of order 1e-1 .. 1)
    print(f"  race screen: 60 repetitions at C2, forward bit-identical, worst gradient difference {worst:.2e} ({worst_k})")


@pytest.mark.parametrize("B,N,Lr,train", [(17, 8, 2, True), (9, 16, 2, False), (10, 13, 3, True), (32, 16, 2, True)])
def test_half_height_tiles_backward_equals_full_height_tiles(B, N, Lr, train):
    """Round 6: the fused backward on 32-row tiles (csrc/dvgr_gatf.h, gatf_half_shape; (32, 16) is C4's per-GPU batch) against the
    64-row form (DVGR_OPT_NO_HALF_GAT_TILES): the same arithmetic per sample, partial sums over samples in another order."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=59, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=59)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 59))
    got = {}
    for half in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, train, True)
        m.half_gat_tiles = half
        _, _, got[half] = _grads(m, inp, R, N, 31)
    worst = max(rel_err(got[True][k], got[False][k]) for k in got[True] if not k.endswith(".bias"))
    assert worst < 5e-5, worst
    # (column sums by float atomics: two runs of ONE form differ by ~2e-5 there; a.bias / fc.bias are zero in exact arithmetic)
    worst_b = max(rel_err(got[True][k], got[False][k]) for k in got[True] if k.endswith(".bias") and not k.endswith((".fc.bias", ".a.bias")))
    assert worst_b < 5e-4, worst_b


def test_half_height_tiles_backward_against_oracle():
    """C4's per-GPU shape (32 QA pairs of 16 clips: 256 half-height workgroups) in training mode against the oracle."""
    B, N, Lr = 32, 16, 2
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=63, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=63)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 63))
    m = _module(N, Lr, "bf16x3", sd_np, True, True)
    seed = 515
    outs, t, _ = _grads(m, inp, R, N, seed)
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, B, N, D, Lr).items()}
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    signs = device_logit_signs(m, L.make_dims(B, N, T, D, WD, Lr), (B, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, masks, lambda o: _scalar(o, R, N), signs)
    what = f"gatb half-height tiles B={B} N={N} L={Lr} bf16x3 train"
    assert_flips_at_kink(stats, what)
    assert_grads(m, t, gin, gw, dbg, Lr, GTOL, what)


@pytest.mark.parametrize("B,N,Lr,train,fused", [(7, 20, 3, True, True), (17, 8, 2, False, True), (6, 24, 2, True, False), (3, 64, 2, True, False),
                                               (33, 20, 2, True, True)])
def test_folded_punish_equals_the_separate_launch(B, N, Lr, train, fused):
    """Round 6: the dX half of a layer's S1 backward (dX += dtau p + the specific GAT's partial; model/utils.py:92-105 backward)
    completed by the NEXT layer's row pass k_sf_bwd (SfBwdArgs::fin_*) against the separate k_punish_bwd launch
    (DVGR_OPT_NO_FOLDED_PUNISH): the same sums in the same order but for dtau's eight head partials (tree against chain).
    L = 3 folds two boundaries; N = 24 / 64 run the two-launch attention backward behind the same row pass."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=67, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=67)
    R = torch.from_numpy(synth.normal("R_gatb", (B, N, D), 1.0, 67))
    got = {}
    for fold in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, train, fused)
        m.folded_punish = fold
        _, _, got[fold] = _grads(m, inp, R, N, 1234)
    worst, wk = 0.0, None
    for k in got[True]:
        if k.endswith((".fc.bias", ".a.bias")):
            continue
        e = rel_err(got[True][k], got[False][k])
        if k.endswith(".bias"):          # column sums by float atomics: two runs of ONE form differ by ~2e-5 there
            assert e < 5e-4, (e, k)
            continue
        if e > worst:
            worst, wk = e, k
    print(f"  folded punish vs separate launch B={B} N={N} L={Lr}: worst rel err {worst:.2e} ({wk})")
    assert worst < 1e-5, (worst, wk)
