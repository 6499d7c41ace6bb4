"""GPU: the fused GAT-view kernel (csrc/dvgr_gatf.hip: projection + logits + softmax + punish-scaled aggregation + ELU + both
dropout sites of one head in ONE launch; model/GraphNN.py:95-111, 174-178) against

  (1) the two-launch form of the same stage (DVGR_OPT_NO_FUSED_GAT: gemm_planes_kernel then k_gat_attn_fwd) -- the projection
      Wh must come out BIT-IDENTICAL (same operand planes, same MFMA order per element), u / v and everything downstream to
      fp32 rounding (the punish gate multiplies the attention column instead of the projection row: (alpha s^T) Wh);
  (2) the oracle, forward and every gradient, at tile shapes the other test files do not reach: a last tile with fewer samples
      than the others, N not a multiple of 4 (reduction steps past the sample's rows), every (MTILES, NT) instantiation,
      single-stream mode (layer 0 fused too), graph_layers = 2.
All through the C ABI via the drop-in module, D = 768 (the kernel's one head width F = 192), bf16x3 (the bench's mode) and
plain bf16."""
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


def _module(N, Lr, dtype, sd_np, train, fused, single=False, G=1, lo8=False, kb64=True):
    m = dv.DualVGRUnit_multiple(word_dim=WD, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=dtype,
                                appearance_graph_layers=G, motion_graph_layers=G)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.cuda().train(train)
    m.keep_saved = True
    m.fused_gat = fused
    m.single_stream = single
    m.lo8_planes = lo8          # True: the fused kernels stream an 8-bit lo plane of their weight operand (dvgr_opts.lo8; 32-deep tiles)
    m.kb64 = kb64               # False: 32-deep K-tiles (DVGR_OPT_KB32)
    return m


def _forward(m, inp, seed):
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    m.forced_seed = seed
    with torch.no_grad():
        outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    torch.cuda.synchronize()
    return outs


def _saved(m, dims, kind, layer, mod, shape):
    off = L.saved_offset(dims, kind, layer, mod)
    n = int(np.prod(shape))
    return m.last_saved[off:off + 4 * n].view(torch.float32).view(shape).cpu().clone()


# (B, N, L): tile shapes.  N = 20 -> 5 row tiles x 4 samples, B = 7 leaves a last tile of 3 samples; N = 8 / 16 / 13 -> 4 row
# tiles; N = 24 / 32 -> 6 row tiles (NT = 2); N = 13, 33 are not multiples of 4; N = 33 / 40 -> NT = 3; N = 64 -> 8 row tiles x 2
# samples (last tile: 1); B = 1 < samples per tile.  Above 128 rows both forms run the projection on bf16 planes (below, the
# two-launch form takes the batch-sized exact-fp32 kernels of dvgr_small.hip: equal to operand rounding only).
SHAPES = [(7, 20, 2), (17, 8, 2), (9, 16, 2), (6, 24, 2), (5, 32, 2), (10, 13, 2), (4, 33, 2), (4, 40, 2), (3, 64, 2), (1, 20, 2)]


@pytest.mark.parametrize("B,N,Lr", SHAPES)
@pytest.mark.parametrize("train", [False, True])
def test_fused_equals_two_launch_form(B, N, Lr, train):
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=41, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=41)
    dims = L.make_dims(B, N, T, D, WD, Lr)
    got = {}
    for fused in (True, False):
        # single stream: layer 0 takes the fused kernel too; bf16 lo planes in both forms (the bit-identity claim is about those:
        # a 64-deep K-tile is two 32-deep steps in the same order, so the MFMA chain per element is the same either way)
        m = _module(N, Lr, "bf16x3", sd_np, train, fused, single=True, lo8=False)
        outs = _forward(m, inp, 777)
        sv = {}
        for l in range(Lr):
            for mi in range(2):
                sv[("Wh", l, mi)] = _saved(m, dims, L.S_WH, l, mi, (B, N, 8 * (D // 4)))
                sv[("u", l, mi)] = _saved(m, dims, L.S_U, l, mi, (B, 8, N))
                sv[("v", l, mi)] = _saved(m, dims, L.S_V, l, mi, (B, 8, N))
        got[fused] = ([outs[0].cpu(), outs[1].cpu(), outs[2].cpu()] + [x.cpu() for lst in outs[3:] for x in lst], sv)
    planes_both = B * N > 128
    if planes_both:    # layer 0's projection sees identical inputs in both forms: bit-identical (same planes, same MFMA chain per element)
        for mi in range(2):
            assert torch.equal(got[True][1][("Wh", 0, mi)], got[False][1][("Wh", 0, mi)]), "layer 0 projection differs"
    worst = 0.0
    for k in got[True][1]:
        worst = max(worst, rel_err(got[True][1][k], got[False][1][k]))
    for a, b in zip(got[True][0], got[False][0]):
        worst = max(worst, rel_err(a, b))
    print(f"  fused vs two launches, B={B} N={N} train={train}: worst rel err {worst:.2e}")
    assert worst < (1e-5 if planes_both else 5e-5), worst
    if train:
        assert float((got[True][0][3] == 0).float().mean()) > 0.1      # the out-site dropout really ran in the fused epilogue


@pytest.mark.parametrize("B,N,Lr,single", [(6, 20, 2, False), (6, 20, 2, True), (5, 13, 2, True), (3, 64, 1, True), (9, 8, 3, False),
                                          (3, 40, 2, True)])
def test_fused_against_oracle_training_mode(B, N, Lr, single):
    """Forward and every gradient against the oracle with the device's dropout masks injected (the kink-aware gate of
    tests/helpers.py), bf16x3."""
    dtype = "bf16x3"
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=43, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=43)
    R = torch.from_numpy(synth.normal("R_gatf", (B, N, D), 1.0, 43))
    m = _module(N, Lr, dtype, sd_np, True, True, single=single)
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    seed = 4242
    m.forced_seed = seed
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])

    def scalar(o, Rr):
        c, d_ = O.aux_losses(o, N)
        return (o[0] * Rr).sum() + 0.3 * ((o[1] * Rr).sum() + (o[2] * Rr).sum()) + c + 1e-8 * d_

    scalar(outs, R.cuda()).backward()
    torch.cuda.synchronize()
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, B, N, D, Lr).items()}
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    signs = device_logit_signs(m, L.make_dims(B, N, T, D, WD, Lr), (B, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, masks, lambda o: scalar(o, R), signs)
    what = f"gatf B={B} N={N} L={Lr} {dtype} single={single}"
    flat_h = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    flat_o = [oo[0], oo[1], oo[2]] + oo[3] + oo[4] + oo[5] + oo[6]
    ferr = max(rel_err(h.detach().cpu(), o) for h, o in zip(flat_h, flat_o))
    print(f"  {what}: forward (all {len(flat_h)} outputs) {ferr:.2e}")
    assert ferr < TOL, ferr
    assert_flips_at_kink(stats, what)
    assert_grads(m, t, gin, gw, dbg, Lr, GTOL, what)


def test_graph_layers_2_fused_equals_two_launch_form():
    """graph_layers = 2: graph layer 1 reads graph layer 0's specific output (its planes live in the list-output slots) and
    the layer's one punish gate; every (layer, graph layer) step takes the fused kernel."""
    B, N, Lr, G = 6, 24, 2, 2
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr, G), seed=44, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=44)
    a = _forward(_module(N, Lr, "bf16x3", sd_np, True, True, single=True, G=G), inp, 9)
    b = _forward(_module(N, Lr, "bf16x3", sd_np, True, False, single=True, G=G), inp, 9)
    fa = [a[0], a[1], a[2]] + [x for lst in a[3:] for x in lst]
    fb = [b[0], b[1], b[2]] + [x for lst in b[3:] for x in lst]
    assert len(fa) == 3 + 4 * Lr * G
    worst = max(rel_err(x.cpu(), y.cpu()) for x, y in zip(fa, fb))
    assert worst < 1e-5, worst


def test_plain_bf16_mode_takes_the_fused_kernel_too():
    """DVGR_BF16 (one plane per operand): layer 0's projection is bit-identical in both forms; downstream, an fp32-rounding
    difference in the attention can flip the bf16 rounding of a later operand element (2^-9 relative), so the outputs are held
    to the mode's own tolerance class, not to fp32 rounding."""
    B, N, Lr = 7, 20, 2
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=45, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=45)
    dims = L.make_dims(B, N, T, D, WD, Lr)
    ma = _module(N, Lr, "bf16", sd_np, True, True, single=True)
    a = _forward(ma, inp, 5)
    wa = [_saved(ma, dims, L.S_WH, 0, mi, (B, N, 8 * (D // 4))) for mi in range(2)]
    mb = _module(N, Lr, "bf16", sd_np, True, False, single=True)
    b = _forward(mb, inp, 5)
    wb = [_saved(mb, dims, L.S_WH, 0, mi, (B, N, 8 * (D // 4))) for mi in range(2)]
    assert torch.equal(wa[0], wb[0]) and torch.equal(wa[1], wb[1])
    worst = max(rel_err(x.cpu(), y.cpu()) for x, y in zip([a[0]] + list(a[3]) + list(a[6]), [b[0]] + list(b[3]) + list(b[6])))
    assert worst < 5e-3, worst


def test_launch_count_drops_by_one_per_fused_layer():
    B, N, Lr = 8, 20, 3
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=47, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=47)
    lib = L.load()
    counts = {}
    for fused in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, True, fused)
        _forward(m, inp, 3)
        n0 = lib.dvgr_dbg_launch_count()
        _forward(m, inp, 3)
        counts[fused] = lib.dvgr_dbg_launch_count() - n0
    # every layer is fused -- round 6: layer 0 too, its share of the query side S0 runs in front of it on the caller's stream
    # (DVGR_OPT_S0_ON_SIDE, the form of rounds 1-5, kept two launches for layer 0); the same flag also selects S6 as one launch
    # (k_mfb_fwd: linear0 / linear1 + ELU + pooling) instead of contraction + pooling kernel
    assert counts[False] - counts[True] == Lr + 1, counts


@pytest.mark.parametrize("B,N,Lr,train", [(9, 20, 3, True), (17, 8, 2, False), (1, 20, 1, True)])
def test_s0_placements_agree(B, N, Lr, train):
    """Where the query side S0 runs (include/dvgr.h: default = layer 0's share on the caller's stream, the rest on the side stream;
    DVGR_OPT_S0_ON_MAIN; DVGR_OPT_S0_ON_SIDE = rounds 1-5) changes streams and the batching of its kernels over the layers, not
    its arithmetic: saved query tensors and gates bit-identical, outputs equal (layer 0 runs the two-launch form under
    DVGR_OPT_S0_ON_SIDE: the attention's reductions in another order)."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=49, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=49)
    dims = L.make_dims(B, N, T, D, WD, Lr)
    got = {}
    for where in ("split", "main", "side"):
        m = _module(N, Lr, "bf16x3", sd_np, train, True)
        m.s0_where = where
        outs = _forward(m, inp, 88)
        gates = [_saved(m, dims, L.S_S, l, mi, (B, N)) for l in range(Lr) for mi in range(2)]
        got[where] = ([outs[0].cpu()] + [x.cpu() for x in outs[3]] + [x.cpu() for x in outs[5]], gates)
    for where in ("main", "side"):
        for i, (a, b) in enumerate(zip(got["split"][1], got[where][1])):
            # layer 0's gates come straight from S0: bit-identical everywhere; the later layers' gates are computed from X', which
            # under "side" went through layer 0's two-launch attention (another reduction order)
            if where == "main" or i < 2:
                assert torch.equal(a, b), (where, i)
            else:
                assert rel_err(a, b) < 5e-6, (where, i)
        worst = max(rel_err(a, b) for a, b in zip(got["split"][0], got[where][0]))
        # ("side": layer 0 in the two-launch form -- below 128 rows its projection runs on the batch-sized exact-fp32 kernels,
        # equal to the bf16x3 planes' operand rounding only)
        assert worst < (1e-12 if where == "main" else 2e-5), (where, worst)


@pytest.mark.parametrize("B,N,Lr,train", [(64, 20, 2, True), (32, 8, 2, False), (7, 20, 2, True)])
def test_s5_logits_from_the_s4_epilogue_equal_its_own_dot_products(B, N, Lr, train):
    """Round 5: the S4 contraction's epilogue hands S5 the logits tanh(z_v P1^T + b) . p2 as partial dots per 32-column block
    (whole 128-row tiles only: B N % 128 == 0 -- the third case keeps S5's own pass); DVGR_OPT_S5_READS_TN is the old form.
    Same values, different summation order: outputs to fp32 rounding."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=61, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=61)
    got = {}
    for reads_tn in (False, True):
        m = _module(N, Lr, "bf16x3", sd_np, train, True)
        m.s5_reads_tn = reads_tn
        outs = _forward(m, inp, 123)
        got[reads_tn] = [outs[0].cpu(), outs[1].cpu(), outs[2].cpu()] + [x.cpu() for lst in outs[3:] for x in lst]
    worst = max(rel_err(a, b) for a, b in zip(got[True], got[False]))
    print(f"  S5 logits from the S4 epilogue vs its own pass, B={B} N={N}: worst rel err {worst:.2e}")
    assert worst < 5e-6, worst


@pytest.mark.parametrize("B,N,Lr,train", [(64, 20, 2, True), (17, 8, 2, False), (7, 20, 3, True)])
def test_8_bit_lo_plane_of_the_weight_operand(B, N, Lr, train):
    """Round 5: the fused kernels stream the lo plane of their weight operand as 8 bits per element + one scale per row
    (dvgr_pack_lo8, csrc/dvgr_gatf.h LO8) instead of bf16.  The lo plane carries bits 9 .. 16 of a weight: 8 bits against the
    row's largest |lo| keep the operand to 2^-17 of the row's largest weight, the grade of the bf16 lo plane.  Forward outputs
    and the saved projection / logit halves against the bf16-lo form (single stream: every layer takes the fused kernel);
    gradients are NOT compared between the two device runs -- a 1e-5 change of a logit next to the LeakyReLU kink flips its
    branch (the oracle tests of this file and of test_hip_gatb.py, which run with the 8-bit plane, hold the gradients with the
    device's branches injected)."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=63, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=63)
    dims = L.make_dims(B, N, T, D, WD, Lr)
    got = {}
    for lo8 in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, train, True, single=True, lo8=lo8, kb64=False)
        outs = _forward(m, inp, 31)
        g = {"out:%d" % i: x.cpu() for i, x in enumerate([outs[0], outs[1], outs[2]] + [y for lst in outs[3:] for y in lst])}
        for l in range(Lr):
            for mi in range(2):
                g["Wh:%d:%d" % (l, mi)] = _saved(m, dims, L.S_WH, l, mi, (B, N, 8 * (D // 4)))
                g["u:%d:%d" % (l, mi)] = _saved(m, dims, L.S_U, l, mi, (B, 8, N))
        got[lo8] = g
    worst, wk = 0.0, None
    for k in got[True]:
        e = rel_err(got[True][k], got[False][k])
        if e > worst:
            worst, wk = e, k
    print(f"  8-bit lo plane vs bf16 lo plane, B={B} N={N} train={train}: worst rel err {worst:.2e} ({wk})")
    assert 0.0 < worst < 5e-5, (worst, wk)          # (> 0: the 8-bit plane really was used)


@pytest.mark.parametrize("B,N,Lr,train", [(64, 20, 2, True), (17, 8, 2, False), (7, 13, 2, True)])
def test_64_deep_k_tiles_equal_32_deep_ones(B, N, Lr, train):
    """Round 5: the fused kernels' K loops on 64-deep K-tiles (whole 128-byte lines per row piece, 2-slot ring) against the
    32-deep tiles (DVGR_OPT_KB32): a 64-deep tile is consumed as two 32-deep MFMA steps in the same order, so every accumulator
    sees the same chain -- outputs and saved projections BIT-identical."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=67, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=67)
    dims = L.make_dims(B, N, T, D, WD, Lr)
    got = {}
    for kb64 in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, train, True, single=True, kb64=kb64)
        outs = _forward(m, inp, 17)
        g = [x.cpu() for x in [outs[0], outs[1], outs[2]] + [y for lst in outs[3:] for y in lst]]
        g += [_saved(m, dims, L.S_WH, l, mi, (B, N, 8 * (D // 4))) for l in range(Lr) for mi in range(2)]
        got[kb64] = g
    for a, b in zip(got[True], got[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,N,Lr,train", [(17, 8, 2, True), (9, 16, 2, False), (10, 13, 2, True), (32, 16, 2, True), (5, 5, 2, True)])
def test_half_height_tiles_equal_full_height_tiles(B, N, Lr, train):
    """Round 6: at N <= 16 a launch whose 64-row tiles would leave most of the chip idle runs 32-row tiles (csrc/dvgr_gatf.h,
    gatf_half_shape -- the shapes above all qualify; (32, 16) is C4's per-GPU batch).  A sample's arithmetic does not depend on
    the tile it sits in: every output equals the 64-row form's (DVGR_OPT_NO_HALF_GAT_TILES) bit for bit."""
    sd_np = synth.xavier_state_dict(O.param_shapes(WD, D, Lr), seed=43, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, WD, seed=43)
    got = {}
    for half in (True, False):
        m = _module(N, Lr, "bf16x3", sd_np, train, True, single=True)
        m.half_gat_tiles = half
        outs = _forward(m, inp, 4242)
        got[half] = [outs[0].cpu(), outs[1].cpu(), outs[2].cpu()] + [x.cpu() for x in outs[3]] + [x.cpu() for x in outs[5]]
    for a, b in zip(got[True], got[False]):
        assert torch.equal(a, b)
