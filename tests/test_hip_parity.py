"""GPU: the HIP path (through the C ABI, via the drop-in nn.Module) against
  (1) the golden vectors recorded from the imported reference (tests/golden/*.npz),
  (2) the oracle (oracle/unit_torch.py, oracle/unit_manual.py) on the same seeded inputs,
  (3) size-independent properties at the full BASELINE.json configuration.
Tolerance (BASELINE.json north_star): 1e-4 relative, fp32 -- max|got - want| / max|want| per
tensor; gradients of the chained graph get 5e-4 (they pass through L dependent layers)."""
import ctypes as C

import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import _lib as L
from dualvgr_videoqa_amd import dropout, synth
from helpers import (REAL_CASES, GoldenCase, aligned_oracle_backward, assert_flips_at_kink, assert_grads,
                     device_logit_signs, rel_err, scalar_loss)
from oracle import unit_manual as MN
from oracle import unit_torch as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
GTOL = 5e-4


def build_module(case: GoldenCase, train=False, dtype="fp32"):
    m = dv.DualVGRUnit_multiple(word_dim=case.Wd, module_dim=case.D, num_of_nodes=case.N, unit_layers=case.L, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)
    m = m.cuda()
    m.train(train)
    m.keep_saved = True
    return m


def run_hip(case: GoldenCase, m, upstream=True):
    t = case.torch_inputs(device="cuda", requires_grad=True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    R = {k: torch.from_numpy(v).cuda() for k, v in case.upstream().items()}
    loss, com, dep = scalar_loss(outs, R, case.N, case.alpha, case.beta)
    loss.backward()
    torch.cuda.synchronize()
    return t, outs, loss, com, dep


def saved_view(m, dims, kind, layer, mod, shape):
    off = L.saved_offset(dims, kind, layer, mod)
    n = int(np.prod(shape))
    return m.last_saved[off:off + 4 * n].view(torch.float32).view(shape).cpu()


def check_stages(case, m, masks=None):
    """Stage-by-stage comparison of the tensors the forward saved against oracle/unit_manual.py."""
    sd = case.torch_state()
    t = case.torch_inputs()
    outs, cache = MN.forward(sd, t["A"], t["M"], t["dq"], t["words"], t["qlen"], case.L, masks=masks)
    dims = L.make_dims(case.B, case.N, case.T, case.D, case.Wd, case.L)
    B, N, T, D, Wd, Fh = case.B, case.N, case.T, case.D, case.Wd, case.D // 4
    report = []
    for l in range(case.L):
        c = cache["layers"][l]
        pairs = [("qa_d", L.S_QA_D, 0, (B, T, D), c["d"]), ("qa_a", L.S_QA_A, 0, (B, T), c["a"]),
                 ("qa_ah", L.S_QA_AH, 0, (B, T), c["ah"]), ("qa_q", L.S_QA_Q, 0, (B, Wd), c["q"])]
        for mi, mod in enumerate(("app", "mot")):
            cm = c[mod]
            pairs += [(f"{mod}.p", L.S_P, mi, (B, D), cm["p"]), (f"{mod}.s", L.S_S, mi, (B, N), cm["s"]),
                      (f"{mod}.Wh", L.S_WH, mi, (B, N, 8 * Fh), cm["Wh"]), (f"{mod}.u", L.S_U, mi, (B, 8, N), cm["u"]),
                      (f"{mod}.v", L.S_V, mi, (B, 8, N), cm["v"]), (f"{mod}.tn", L.S_TN, mi, (2, B, N, D), cm["tn"].transpose(0, 1)),
                      (f"{mod}.beta", L.S_BETA, mi, (B, N, 2), cm["beta"].permute(0, 2, 1)),
                      (f"{mod}.x_out", L.S_X, mi, (B, N, D), cm["x"] + cm["o"])]
        for name, kind, mi, shape, want in pairs:
            got = saved_view(m, dims, kind, l, mi, shape)
            report.append((f"L{l}.{name}", rel_err(got, want)))
    for name, kind, shape, want in (("mfb.x0", L.S_MFB_X0, (B, N, 512), cache["x0"]), ("mfb.x1", L.S_MFB_X1, (B, N, 512), cache["x1"]),
                                    ("mfb.z", L.S_MFB_Z, (B, N, 256), cache["zf"])):
        report.append((name, rel_err(saved_view(m, dims, kind, 0, 0, shape), want)))
    return report


def assert_report(report, tol):
    bad = [(n, e) for n, e in report if not (e < tol)]
    for n, e in report:
        print(f"    stage {n:24s} rel_err {e:.3e}")
    assert not bad, f"stages off: {bad}"


@pytest.mark.parametrize("name", ["g_small_eval", "g_quirk_n1"])
def test_forward_stages_small(name):
    case = GoldenCase(name)
    m = build_module(case)
    with torch.no_grad():
        t = case.torch_inputs(device="cuda")
        m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    torch.cuda.synchronize()
    assert_report(check_stages(case, m), TOL)


@pytest.mark.parametrize("name", ["g_small_eval", "g_quirk_n1"])
def test_small_golden_forward_backward(name):
    case = GoldenCase(name)
    z = case.z
    m = build_module(case)
    t, outs, loss, com, dep = run_hip(case, m)
    vis, aqe, mqe, ca, cm, aq, mq = outs
    errs = {"vis": rel_err(vis.detach().cpu(), z["out/vis"]), "aq_embed": rel_err(aqe.detach().cpu(), z["out/aq_embed"]),
            "mq_embed": rel_err(mqe.detach().cpu(), z["out/mq_embed"])}
    for l in range(case.L):
        errs[f"com_app{l}"] = rel_err(ca[l].detach().cpu(), z[f"out/com_app{l}"])
        errs[f"com_motion{l}"] = rel_err(cm[l].detach().cpu(), z[f"out/com_motion{l}"])
        errs[f"aq_fusion{l}"] = rel_err(aq[l].detach().cpu(), z[f"out/aq_fusion{l}"])
        errs[f"mq_fusion{l}"] = rel_err(mq[l].detach().cpu(), z[f"out/mq_fusion{l}"])
    print("  forward:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < TOL, errs
    assert abs(float(loss) - float(z["loss"])) < 1e-3 * max(1.0, abs(float(z["loss"])))
    gerrs = {k: rel_err(t[k].grad.cpu(), z["gin/" + k]) for k in ("A", "M", "dq", "words")}
    print("  input grads:", {k: f"{v:.2e}" for k, v in gerrs.items()})
    big = set(z["gwbig_names"].tolist()) if "gwbig_names" in z.files else set()
    werrs = {}
    for k, p in m.named_parameters():
        g = p.grad.cpu().numpy()
        if k.endswith(".fc.bias"):
            assert float(np.abs(g).max()) < 1e-4          # analytically zero
            continue
        if k in big:
            i = sorted(big).index(k)
            werrs[k] = float(np.abs(g.ravel()[:64] - z["gwbig_head"][i]).max()) / (float(np.abs(z["gwbig_head"][i]).max()) + 1e-2)
        else:
            want = z["gw/" + k]
            werrs[k] = float(np.abs(g - want).max()) / max(float(np.abs(want).max()), 1e-3)
    worst = sorted(werrs.items(), key=lambda kv: -kv[1])[:8]
    print("  worst weight grads:", [(k, f"{v:.2e}") for k, v in worst])
    assert max(gerrs.values()) < GTOL, gerrs
    assert worst[0][1] < GTOL, worst


@pytest.mark.parametrize("name", REAL_CASES)
def test_real_dim_golden(name):
    case = GoldenCase(name)
    z = case.z
    m = build_module(case)
    t, outs, loss, com, dep = run_hip(case, m)
    vis, aqe, mqe, ca, cm, aq, mq = outs
    last = case.L - 1
    errs = {"vis": rel_err(vis.detach().cpu(), z["out/vis"]), "mq_embed": rel_err(mqe.detach().cpu(), z["out/mq_embed"]),
            "com_app": rel_err(ca[last].detach().cpu(), z[f"out/com_app{last}"]),
            "aq_fusion": rel_err(aq[last].detach().cpu(), z[f"out/aq_fusion{last}"])}
    print("  forward:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert_report(check_stages(case, m), TOL)
    assert max(errs.values()) < TOL, errs
    gerrs = {k: rel_err(t[k].grad.cpu(), z["gin/" + k]) for k in ("A", "M", "dq", "words")}
    print("  input grads:", {k: f"{v:.2e}" for k, v in gerrs.items()})
    assert max(gerrs.values()) < GTOL, gerrs
    named = dict(m.named_parameters())
    werrs = {}
    for i, k in enumerate(z["gw_names"].tolist()):
        g = named[k].grad.double().cpu()
        if k.endswith(".fc.bias"):
            continue
        werrs[k] = max(abs(float(g.sum()) - z["gw_sum"][i]) / max(z["gw_abs"][i], 1e-2),
                       float(np.abs(np.resize(g.flatten()[:8].numpy(), 8) - z["gw_head"][i]).max()) / (float(np.abs(z["gw_head"][i]).max()) + 1e-2))
    worst = sorted(werrs.items(), key=lambda kv: -kv[1])[:8]
    print("  worst weight grads:", [(k, f"{v:.2e}") for k, v in worst])
    assert worst[0][1] < GTOL, worst


def test_train_mode_mask_injected_parity():
    """Training mode: the HIP forward draws its keep-masks from the counter RNG; the numpy replica
    regenerates the same masks, which are injected into the oracle -> exact-parity check of all 24
    dropout sites per layer, forward and backward."""
    case = GoldenCase("g_small_eval")
    m = build_module(case, train=True)
    seed = 0x0123_4567_89AB_CDEF
    m.forced_seed = seed
    t, outs, loss, _, _ = run_hip(case, m)
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, case.B, case.N, case.D, case.L).items()}
    assert_report(check_stages(case, m, masks=masks), TOL)
    sd = {k: v.requires_grad_(True) for k, v in case.torch_state().items()}
    ti = case.torch_inputs(requires_grad=True)
    oouts = O.unit_forward(sd, ti["A"], ti["M"], ti["dq"], ti["words"], ti["qlen"], case.L, 1, masks=masks, naive=False)
    R = {k: torch.from_numpy(v) for k, v in case.upstream().items()}
    oloss, _, _ = scalar_loss(oouts, R, case.N, case.alpha, case.beta)
    oloss.backward()
    flat_h = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    flat_o = [oouts[0], oouts[1], oouts[2]] + oouts[3] + oouts[4] + oouts[5] + oouts[6]
    errs = [rel_err(h.detach().cpu(), o.detach()) for h, o in zip(flat_h, flat_o)]
    print("  forward errs:", [f"{e:.2e}" for e in errs])
    assert max(errs) < TOL
    # some outputs must actually be dropped (15 %), i.e. the train path really ran
    frac0 = float((outs[3][0] == 0).float().mean())
    assert 0.10 < frac0 < 0.20, frac0
    gerrs = {k: rel_err(t[k].grad.cpu(), ti[k].grad) for k in ("A", "M", "dq", "words")}
    print("  input grads:", {k: f"{v:.2e}" for k, v in gerrs.items()})
    assert max(gerrs.values()) < GTOL
    werrs = {}
    for k, p in m.named_parameters():
        if k.endswith(".fc.bias"):
            continue
        want = sd[k].grad
        werrs[k] = float((p.grad.cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-3)
    worst = sorted(werrs.items(), key=lambda kv: -kv[1])[:6]
    print("  worst weight grads:", [(k, f"{v:.2e}") for k, v in worst])
    assert worst[0][1] < GTOL, worst


def test_train_mode_statistics_and_determinism():
    case = GoldenCase("g_real_n20_l1")
    m = build_module(case, train=True)
    t = case.torch_inputs(device="cuda")
    with torch.no_grad():
        m.forced_seed = 11
        a = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        b = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        m.forced_seed = 12
        c = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    assert torch.equal(a[0], b[0]) and torch.equal(a[5][0], b[5][0])      # same seed -> same masks
    assert not torch.equal(a[5][0], c[5][0])
    drop = float((a[5][0] == 0).float().mean())                             # output-site dropout rate
    assert abs(drop - 0.15) < 0.01, drop


def _full_config(B=64, N=20, T=15, Lr=3, D=768, Wd=300, seed=666, dtype="fp32"):
    shapes = O.param_shapes(Wd, D, Lr)
    sd_np = synth.xavier_state_dict(shapes, seed=seed, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=seed)
    m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m.keep_saved = True
    return m.cuda().eval(), sd_np, inp


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_full_size_against_oracle_and_properties(dtype):
    """BASELINE.json configs[1] shape (B=64, N=20, L=3, D=768), in exact fp32 AND in the benchmark's default operand
    mode (bf16x3), same gates: forward + backward against the oracle, then size-independent properties:
    batch-permutation equivariance, bitwise run-to-run determinism."""
    m, sd_np, inp = _full_config(dtype=dtype)
    B, N, D = 64, 20, 768
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    Rv = torch.from_numpy(synth.normal("Rfull", (B, N, D), 1.0)).cuda()
    com, dep = O.aux_losses(outs, N)
    loss = (outs[0] * Rv).sum() + com + 1e-8 * dep
    loss.backward()
    torch.cuda.synchronize()
    # oracle on the host (autograd form for the forward and the loss; hand-derived backward with aligned LeakyReLU branches)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}

    def upstream(o):
        c, d_ = O.aux_losses(o, N)
        return (o[0] * Rv.cpu()).sum() + c + 1e-8 * d_

    signs = device_logit_signs(m, L.make_dims(B, N, 15, D, 300, 3), (B, N, 3))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, 3, None, upstream, signs)
    errs = {"vis": rel_err(outs[0].detach().cpu(), oo[0]), "aq_embed": rel_err(outs[1].detach().cpu(), oo[1]),
            "com_app2": rel_err(outs[3][2].detach().cpu(), oo[3][2]), "mq_fusion2": rel_err(outs[6][2].detach().cpu(), oo[6][2])}
    print(f"  {dtype} forward:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < TOL, errs
    assert_flips_at_kink(stats, f"C2 {dtype}")
    assert_grads(m, t, gin, gw, dbg, 3, GTOL, f"C2 {dtype}")
    # ---- properties ---------------------------------------------------------------------------
    with torch.no_grad():
        base = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        again = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
        assert all(torch.equal(x, y) for x, y in zip([base[0], base[1], base[2]] + base[3] + base[5],
                                                     [again[0], again[1], again[2]] + again[3] + again[5])), "not deterministic"
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
        pouts = m(t["A"][perm], t["M"][perm], t["dq"][perm], t["words"][perm], t["qlen"][perm])
        assert float((pouts[0] - base[0][perm]).abs().max()) < 1e-5 * float(base[0].abs().max())   # samples are independent
        # growing the padded question length T with zero rows changes nothing but the softmax mass on padding:
        # the reference renormalises after masking, so only rows t < len matter up to the 1e-5 epsilon
        assert torch.isfinite(base[0]).all()


def test_list_outputs_stay_on_device_and_grad_accepts_none():
    case = GoldenCase("g_small_eval")
    m = build_module(case)
    t = case.torch_inputs(device="cuda", requires_grad=True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    assert all(o.is_cuda for o in outs[3] + outs[4] + outs[5] + outs[6])
    assert len(outs[3]) == case.L and outs[3][0].shape == (case.B, case.N, case.D)
    outs[0].sum().backward()                      # only visual_feature has an upstream gradient
    torch.cuda.synchronize()
    assert torch.isfinite(t["A"].grad).all() and float(t["A"].grad.abs().sum()) > 0


# ---- reduced-precision MFMA operand modes ----------------------------------------------------------
@pytest.mark.parametrize("name", ["g_real_n20_l3", "g_real_n8_l3"])
def test_bf16x3_meets_the_fp32_parity_gate(name):
    """DVGR_BF16X3 (every operand split hi + lo in bf16, three MFMAs per product, fp32 accumulate) is held
    to the SAME tolerances as the exact-fp32 path: 1e-4 relative forward, 5e-4 gradients."""
    case = GoldenCase(name)
    z = case.z
    m = build_module(case, dtype="bf16x3")
    t, outs, loss, com, dep = run_hip(case, m)
    vis, aqe, mqe, ca, cm, aq, mq = outs
    last = case.L - 1
    errs = {"vis": rel_err(vis.detach().cpu(), z["out/vis"]), "mq_embed": rel_err(mqe.detach().cpu(), z["out/mq_embed"]),
            "com_app": rel_err(ca[last].detach().cpu(), z[f"out/com_app{last}"]),
            "aq_fusion": rel_err(aq[last].detach().cpu(), z[f"out/aq_fusion{last}"])}
    print("  bf16x3 forward:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < TOL, errs
    gerrs = {k: rel_err(t[k].grad.cpu(), z["gin/" + k]) for k in ("A", "M", "dq", "words")}
    print("  bf16x3 input grads:", {k: f"{v:.2e}" for k, v in gerrs.items()})
    assert max(gerrs.values()) < GTOL, gerrs
    named = dict(m.named_parameters())
    werrs = {}
    for i, k in enumerate(z["gw_names"].tolist()):
        if k.endswith(".fc.bias"):
            continue
        g = named[k].grad.double().cpu()
        werrs[k] = max(abs(float(g.sum()) - z["gw_sum"][i]) / max(z["gw_abs"][i], 1e-2),
                       float(np.abs(np.resize(g.flatten()[:8].numpy(), 8) - z["gw_head"][i]).max()) / (float(np.abs(z["gw_head"][i]).max()) + 1e-2))
    worst = sorted(werrs.items(), key=lambda kv: -kv[1])[:6]
    print("  bf16x3 worst weight grads:", [(k, f"{v:.2e}") for k, v in worst])
    # the scalar a.bias gradients are sums of B*N*N signed terms that cancel to ~1e-3 of their magnitude: against the
    # golden (plain metric) they are only sanity-checked here; test_other_graph_sizes / test_full_size / test_hip_configs
    # measure them against their cancellation scale (helpers.assert_grads, 1e-5 of the sum of absolute terms)
    assert max(v for k, v in werrs.items() if not k.endswith(".a.bias")) < GTOL, worst
    assert max(v for k, v in werrs.items() if k.endswith(".a.bias")) < 4 * GTOL, worst


def test_bf16_operands_stated_tolerance():
    """DVGR_BF16 (operands rounded once to bf16, fp32 accumulate): 8 significant bits per operand, so the
    stated tolerance is 3e-2 relative forward / 1e-1 on input gradients; argmax-level agreement is what this
    mode is for, not the 1e-4 gate."""
    case = GoldenCase("g_real_n20_l3")
    z = case.z
    m = build_module(case, dtype="bf16")
    t, outs, loss, com, dep = run_hip(case, m)
    errs = {"vis": rel_err(outs[0].detach().cpu(), z["out/vis"]), "mq_embed": rel_err(outs[2].detach().cpu(), z["out/mq_embed"])}
    gerrs = {k: rel_err(t[k].grad.cpu(), z["gin/" + k]) for k in ("A", "M", "dq", "words")}
    print("  bf16 forward:", {k: f"{v:.2e}" for k, v in errs.items()}, "grads:", {k: f"{v:.2e}" for k, v in gerrs.items()})
    assert max(errs.values()) < 3e-2 and max(gerrs.values()) < 1e-1


@pytest.mark.parametrize("N,D,Lr,B,T", [(64, 128, 2, 3, 7), (16, 256, 6, 4, 5), (8, 768, 3, 5, 12), (33, 64, 1, 2, 3),
                                        (20, 768, 1, 1, 1), (64, 768, 1, 2, 40), (2, 64, 2, 130, 2),
                                        # every fragment-edge case of the MFMA attention kernels (16-row tiles, 4-wide k-steps)
                                        (3, 64, 1, 2, 3), (15, 64, 1, 2, 3), (17, 64, 2, 2, 3), (31, 64, 1, 2, 3), (32, 64, 1, 2, 3),
                                        (47, 64, 1, 2, 3), (48, 64, 2, 2, 3), (49, 64, 1, 2, 3), (63, 128, 1, 2, 3)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_other_graph_sizes_against_oracle(N, D, Lr, B, T, dtype):
    """Clip counts / depths of the other BASELINE.json configs (N = 8, 16, 64; L up to 6), an odd N, and the corners
    (one sample with a one-word question; the largest graph at full width with a long question; a batch wider than
    one 128-row tile of two-clip graphs), forward + backward against the oracle on the same seeded inputs.  Both operand
    modes are held to the SAME gates (1e-4 forward, 5e-4 gradients); the discontinuous LeakyReLU derivative is handled
    by counting the logits that sit on the other side of the kink and aligning the oracle's branch on exactly those
    (helpers.aligned_oracle_backward)."""
    Wd = 300 if D == 768 else 32
    sd_np = synth.xavier_state_dict(O.param_shapes(Wd, D, Lr), seed=11, bias_range=0.1)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=11)
    m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.cuda().eval()
    m.keep_saved = True
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    Rv = torch.from_numpy(synth.normal("Rv", (B, N, D), 1.0, 11))

    def upstream(o, R):
        return (o[0] * R).sum() + sum((x * R).sum() for x in list(o[3]) + list(o[6]))

    upstream(outs, Rv.cuda()).backward()
    torch.cuda.synchronize()
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    signs = device_logit_signs(m, L.make_dims(B, N, T, D, Wd, Lr), (B, N, Lr))
    oo, gin, gw, dbg, stats = aligned_oracle_backward(sd, ti, Lr, None, lambda o: upstream(o, Rv), signs)
    errs = [rel_err(outs[0].detach().cpu(), oo[0]), rel_err(outs[2].detach().cpu(), oo[2]),
            rel_err(outs[4][-1].detach().cpu(), oo[4][-1])]
    print(f"  N={N} D={D} L={Lr} {dtype}: fwd {max(errs):.2e}")
    assert max(errs) < TOL
    assert_flips_at_kink(stats, f"N={N} D={D} L={Lr} {dtype}")
    assert_grads(m, t, gin, gw, dbg, Lr, GTOL, f"N={N} D={D} L={Lr} {dtype}")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("train", [False, True])
def test_side_stream_schedule_equals_single_stream(train, fused):
    """The two-stream schedules (forward: query chain + loss on the side stream; backward: weight gradients and the query
    branch) against DVGR_OPT_SINGLE_STREAM on the full-size problem, several times in a row: any missing dependency edge
    shows up as a mismatch.  Tolerance covers only the order of float atomics in split-K / column-sum accumulation."""
    B, N, Lr, T, D, Wd = 64, 20, 3, 15, 768, 300
    shapes = O.param_shapes(Wd, D, Lr, 1)
    sd = synth.xavier_state_dict(shapes, seed=5, bias_range=0.1)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=5)
    results = []
    for single in (True, False, False, False):
        m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        m = m.cuda().train(train)
        m.forced_seed = 4242
        m.single_stream = single
        # fused = False: both schedules launch the same kernels, only the order of float atomics may differ (2e-6).  fused =
        # True (the default): the single-stream schedule takes the fused GAT-view kernel at layer 0 as well, the two-stream
        # one keeps two launches there (csrc/dvgr_api.hip) -- equal to fp32 rounding of the attention's operand order (1e-5).
        m.fused_gat = fused
        m.loss_in_forward = (1.0, 1e-3, 1.0)
        t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
        for k in ("A", "M", "dq", "words"):
            t[k].requires_grad_(True)
        for _ in range(3):                      # back-to-back steps: the schedules also have to be safe across calls
            m.zero_grad_flat()
            for k in ("A", "M", "dq", "words"):
                t[k].grad = None
            raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
            terms, g = m.last_loss
            torch.autograd.backward([raw[0], *raw[3:]], [g[0], *g[3:]])
        torch.cuda.synchronize()
        results.append(([r.detach().cpu().numpy() for r in raw] + [terms.cpu().numpy()] +
                        [t[k].grad.cpu().numpy() for k in ("A", "M", "dq", "words")] + [m.last_flat_grad.cpu().numpy()]))
    ref = results[0]
    for other in results[1:]:
        for a, b in zip(ref, other):
            assert rel_err(b, a) < (1e-5 if fused else 2e-6)


@pytest.mark.parametrize("N,D,Lr,B,T", [(20, 128, 3, 3, 6), (33, 64, 2, 2, 4), (64, 64, 1, 2, 3), (16, 256, 2, 70, 5)])
def test_train_mode_mask_injected_other_sizes(N, D, Lr, B, T):
    """Training mode beyond the small golden case: all dropout sites, the S5 kernel that also prepares the next layer's
    gate and dropped copies, the one-launch S2 dX with its scratch partial -- against the oracle fed with the same masks."""
    Wd = 32
    sd_np = synth.xavier_state_dict(O.param_shapes(Wd, D, Lr), seed=21, bias_range=0.1)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=21)
    m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="fp32")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.cuda().train()
    seed = 987654321
    m.forced_seed = seed
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    Rv = torch.from_numpy(synth.normal("Rv", (B, N, D), 1.0, 21)).cuda()
    (outs[0] * Rv).sum().add(sum((x * Rv).sum() for x in outs[3] + outs[4] + outs[5] + outs[6])).backward()
    torch.cuda.synchronize()
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(seed, B, N, D, Lr).items()}
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        ti[k].requires_grad_(True)
    oo = O.unit_forward(sd, ti["A"], ti["M"], ti["dq"], ti["words"], ti["qlen"], Lr, 1, masks=masks, naive=False)
    (oo[0] * Rv.cpu()).sum().add(sum((x * Rv.cpu()).sum() for x in oo[3] + oo[4] + oo[5] + oo[6])).backward()
    flat_h = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    flat_o = [oo[0], oo[1], oo[2]] + oo[3] + oo[4] + oo[5] + oo[6]
    assert max(rel_err(h.detach().cpu(), o.detach()) for h, o in zip(flat_h, flat_o)) < TOL
    assert max(rel_err(t[k].grad.cpu(), ti[k].grad) for k in ("A", "M", "dq", "words")) < GTOL
    werr = max(float((p.grad.cpu() - sd[k].grad).abs().max()) / max(float(sd[k].grad.abs().max()), 1e-3)
               for k, p in m.named_parameters() if not k.endswith(".fc.bias") and not k.endswith(".a.bias"))
    assert werr < GTOL, werr


# ---- graph_layers > 1 (model/models.py:150-160: GAT module [i + j], shared between unit layers) ------------------------------
def _build_g(case, train=False, dtype="fp32"):
    m = dv.DualVGRUnit_multiple(word_dim=case.Wd, module_dim=case.D, num_of_nodes=case.N, appearance_graph_layers=case.G,
                                motion_graph_layers=case.G, unit_layers=case.L, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)   # all L * G modules per family
    m = m.cuda()
    m.train(train)
    m.keep_saved = True
    return m


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_graph_layers_2_golden_forward_backward(dtype):
    """tests/golden/g_quirk_g2.npz: the imported reference with graph_layers = 2 (its `[i + j]` indexing makes unit layers
    share GAT modules and leaves the last modules unused).  Every list entry (L * G per list), the loss, every input and
    parameter gradient -- shared modules accumulate over the pairs that use them, unused ones get exact zeros."""
    case = GoldenCase("g_quirk_g2")
    assert case.G == 2
    z = case.z
    m = _build_g(case, dtype=dtype)
    t, outs, loss, com, dep = run_hip(case, m)
    vis, aqe, mqe, ca, cm, aq, mq = outs
    assert len(ca) == len(cm) == len(aq) == len(mq) == case.L * case.G
    errs = {"vis": rel_err(vis.detach().cpu(), z["out/vis"]), "aq_embed": rel_err(aqe.detach().cpu(), z["out/aq_embed"]),
            "mq_embed": rel_err(mqe.detach().cpu(), z["out/mq_embed"])}
    for e in range(case.L * case.G):
        errs[f"com_app{e}"] = rel_err(ca[e].detach().cpu(), z[f"out/com_app{e}"])
        errs[f"com_motion{e}"] = rel_err(cm[e].detach().cpu(), z[f"out/com_motion{e}"])
        errs[f"aq_fusion{e}"] = rel_err(aq[e].detach().cpu(), z[f"out/aq_fusion{e}"])
        errs[f"mq_fusion{e}"] = rel_err(mq[e].detach().cpu(), z[f"out/mq_fusion{e}"])
    print("  forward:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < TOL, errs
    assert abs(float(loss) - float(z["loss"])) < 1e-3 * max(1.0, abs(float(z["loss"])))
    gerrs = {k: rel_err(t[k].grad.cpu(), z["gin/" + k]) for k in ("A", "M", "dq", "words")}
    print("  input grads:", {k: f"{v:.2e}" for k, v in gerrs.items()})
    assert max(gerrs.values()) < GTOL, gerrs
    werrs, unused = {}, 0
    last_used = case.L + case.G - 2
    big = sorted(set(z["gwbig_names"].tolist())) if "gwbig_names" in z.files else []
    for k, p in m.named_parameters():
        g = p.grad.cpu().numpy()
        if k.endswith(".fc.bias"):
            continue
        if k in big:                                                   # large matrices: their first 64 elements are recorded
            head = z["gwbig_head"][big.index(k)]
            werrs[k] = float(np.abs(g.ravel()[:64] - head).max()) / (float(np.abs(head).max()) + 1e-2)
            continue
        want = z["gw/" + k]
        fam = k.split(".")[0]
        if fam in ("acGCN", "appearance_GCN", "mcGCN", "motion_GCN") and int(k.split(".")[1]) > last_used:
            assert float(np.abs(g).max()) == 0.0 and float(np.abs(want).max()) == 0.0, k     # never touched by the reference
            unused += 1
            continue
        werrs[k] = float(np.abs(g - want).max()) / max(float(np.abs(want).max()), 1e-3)
    worst = sorted(werrs.items(), key=lambda kv: -kv[1])[:6]
    print("  worst weight grads:", [(k, f"{v:.2e}") for k, v in worst], "unused tensors:", unused)
    assert unused > 0
    assert worst[0][1] < (GTOL if dtype == "fp32" else 4 * GTOL), worst


@pytest.mark.parametrize("G", [2, 3])
def test_graph_layers_training_mode_against_the_oracle(G):
    """Training mode with G graph layers: the device's keep-masks of every (unit layer, graph layer) site injected into
    the autograd oracle (oracle/unit_torch.py restates the [i + j] loop), forward and every gradient."""
    Wd, D, N, Lr, B, T = 32, 64, 6, 3, 5, 7
    sd_np = synth.xavier_state_dict(O.param_shapes(Wd, D, Lr, G), seed=17, bias_range=0.1)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=17)
    m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, appearance_graph_layers=G, motion_graph_layers=G,
                                unit_layers=Lr, dtype="fp32")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    m = m.cuda().train()
    m.forced_seed = 424242
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        t[k].requires_grad_(True)
    outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    R = torch.from_numpy(synth.normal("Rg", (B, N, D), 1.0, 17))

    def upstream(o, Rr):
        c, d_ = O.aux_losses(o, N)
        return (o[0] * Rr).sum() + 0.3 * (o[1] * Rr).sum() + sum(((x * Rr).sum() * (0.1 + 0.05 * i)) for i, x in enumerate(o[3] + o[5])) \
            + sum((x * Rr).sum() * 0.07 for x in o[4] + o[6]) + c + 1e-3 * d_

    upstream(outs, R.cuda()).backward()
    torch.cuda.synchronize()
    from dualvgr_videoqa_amd import dropout
    masks = {k: torch.from_numpy(v) for k, v in dropout.site_masks(424242, B, N, D, Lr, G).items()}
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
    ti = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    for k in ("A", "M", "dq", "words"):
        ti[k].requires_grad_(True)
    oo = O.unit_forward(sd, ti["A"], ti["M"], ti["dq"], ti["words"], ti["qlen"], Lr, G, masks=masks, naive=False)
    upstream(oo, R).backward()
    flat_h = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    flat_o = [oo[0], oo[1], oo[2]] + list(oo[3]) + list(oo[4]) + list(oo[5]) + list(oo[6])
    assert len(flat_h) == len(flat_o) == 3 + 4 * Lr * G
    ferr = max(rel_err(h.detach().cpu(), o.detach()) for h, o in zip(flat_h, flat_o))
    gerr = max(rel_err(t[k].grad.cpu(), ti[k].grad) for k in ("A", "M", "dq", "words"))
    werrs = {}
    for k, p in m.named_parameters():
        if k.endswith(".fc.bias"):
            continue
        want = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        werrs[k] = float((p.grad.cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-3)
    worst = sorted(werrs.items(), key=lambda kv: -kv[1])[:5]
    print(f"  G={G}: forward {ferr:.2e}, input grads {gerr:.2e}, worst weight grads {[(k, f'{v:.2e}') for k, v in worst]}")
    assert ferr < TOL and gerr < GTOL and worst[0][1] < GTOL, (ferr, gerr, worst)
