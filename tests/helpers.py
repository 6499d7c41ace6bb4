"""Shared helpers for the parity tests: golden-case loading and error metrics."""
import os

import numpy as np
import torch

from dualvgr_videoqa_amd import synth
from oracle import unit_torch as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_CASES = ["g_small_eval", "g_small_train", "g_quirk_n1", "g_quirk_g2"]
REAL_CASES = ["g_real_n20_l1", "g_real_n20_l3", "g_real_n8_l3", "g_real_n16_l1"]


class GoldenCase:
    """One .npz written by tests/golden/make_golden.py, with weights/inputs regenerated from
    the (seed, name) generator exactly as that script built them."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        (self.Wd, self.D, self.N, self.L, self.B, self.T, self.G, train, self.seed) = [int(v) for v in self.z["meta"]]
        self.train = bool(train)
        self.alpha, self.beta, self.bias_range = [float(v) for v in self.z["meta_f"]]
        self.full = "in/A" in self.z.files
        shapes = O.param_shapes(self.Wd, self.D, self.L, self.G)
        self.sd_np = synth.xavier_state_dict(shapes, seed=self.seed, bias_range=self.bias_range)
        chk = np.array([float(np.abs(self.sd_np[k].astype(np.float64)).sum()) for k in sorted(self.sd_np)])
        assert np.allclose(chk, self.z["w_checksum"], rtol=1e-12), "weight generator drifted"
        inp = synth.unit_inputs(self.B, self.N, self.T, self.D, self.Wd, seed=self.seed)
        qlen = self.z["qlen"].astype(np.int64)
        if not np.array_equal(qlen, inp["qlen"]):
            valid = (np.arange(self.T)[None, :] < qlen[:, None]).astype(np.float32)
            inp["dq"] = synth.normal("dq", (self.B, self.T, self.D), 0.5, self.seed) * valid[:, :, None]
            inp["qlen"] = qlen
        self.inp = inp
        if self.full:
            for k in ("A", "M", "dq", "words"):
                assert np.array_equal(self.inp[k], self.z["in/" + k]), "input generator drifted: " + k

    # upstream-gradient weights, as make_golden.upstream_weights
    def upstream(self):
        r = {"R_vis": synth.normal("R_vis", (self.B, self.N, self.D), 1.0, self.seed),
             "R_aq": synth.normal("R_aq", (self.B, self.N, self.D), 0.3, self.seed),
             "R_mq": synth.normal("R_mq", (self.B, self.N, self.D), 0.3, self.seed)}
        for l in range(self.L * self.G):
            for nm in ("ca", "cm", "aq", "mq"):
                r[f"R_{nm}{l}"] = synth.normal(f"R_{nm}{l}", (self.B, self.N, self.D), 0.2, self.seed)
        return r

    def masks(self):
        if not self.train:
            return None
        out = {}
        for site, shape in O.dropout_site_shapes(self.B, self.N, self.D, self.L, self.G).items():
            bits = np.unpackbits(self.z["mask/" + site])[: int(np.prod(shape))]
            out[site] = torch.from_numpy(bits.reshape(shape).astype(np.float32))
        return out

    def torch_state(self, device="cpu", dtype=torch.float32):
        return {k: torch.from_numpy(v.copy()).to(device=device, dtype=dtype) for k, v in self.sd_np.items()}

    def torch_inputs(self, device="cpu", dtype=torch.float32, requires_grad=False):
        t = {}
        for k in ("A", "M", "dq", "words"):
            t[k] = torch.from_numpy(self.inp[k].copy()).to(device=device, dtype=dtype).requires_grad_(requires_grad)
        t["qlen"] = torch.from_numpy(self.inp["qlen"].copy()).to(device)
        return t


def scalar_loss(outs, R, N, alpha, beta, common_loss=O.common_loss, loss_dependence=O.loss_dependence):
    """The scalar the golden gradients were taken of (make_golden.scalar_loss)."""
    vis, aqe, mqe, ca, cm, aq, mq = outs
    n = len(aq)
    loss = (vis * R["R_vis"]).sum() + (aqe * R["R_aq"]).sum() + (mqe * R["R_mq"]).sum()
    com = 0
    dep = 0
    for l in range(n):
        com = com + common_loss(ca[l], cm[l])
        dep = dep + loss_dependence(aq[l], ca[l], N) + loss_dependence(mq[l], cm[l], N)
        loss = loss + (ca[l] * R[f"R_ca{l}"]).sum() + (cm[l] * R[f"R_cm{l}"]).sum()
        loss = loss + (aq[l] * R[f"R_aq{l}"]).sum() + (mq[l] * R[f"R_mq{l}"]).sum()
    return loss + alpha * com / n + beta * dep / n, com / n, dep / n


def rel_err(got, want):
    """max |got - want| / max |want| (the 'relative fp32' metric of BASELINE.json's north_star)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    denom = max(float(np.abs(want).max()), 1e-30)
    return float(np.abs(got - want).max()) / denom


# ---- LeakyReLU-kink-aware gradient oracle --------------------------------------------------------------------------------
# d LeakyReLU(pre)/d pre jumps at pre = 0.  An implementation whose attention logits differ from the oracle's by rounding
# (any dtype: ~1e-7 relative in exact fp32, ~1e-5 in bf16x3) can sit on the other side of the kink for a few of the
# B * 8 * N * N * L * 2 logits, and each such entry moves gradients by O(entry), far above any rounding tolerance.  The
# parity tests therefore (1) read the logits the DEVICE used (saved u, v + the a.bias parameter), (2) count the entries
# whose sign differs from the oracle's and require every one of them to lie at the kink (|pre| tiny against the logit
# scale), (3) evaluate the hand-derived oracle backward with the device's branch on exactly those entries, and hold every
# gradient to the plain 5e-4 gate.
def device_logit_signs(m, dims, case_dims):
    """{(layer, 'app'|'mot'): (pre > 0 as the device saw it, pre)} from the forward's saved u, v and the a.bias parameters.
    `m.keep_saved` must have been True during the forward."""
    from dualvgr_videoqa_amd import _lib as L
    B, N, Lr = case_dims
    out = {}
    named = dict(m.named_parameters())
    for l in range(Lr):
        for mi, (mod, fams) in enumerate((("app", ("acGCN", "appearance_GCN")), ("mot", ("mcGCN", "motion_GCN")))):
            uv = []
            for kind in (L.S_U, L.S_V):
                off = L.saved_offset(dims, kind, l, mi)
                uv.append(m.last_saved[off:off + 4 * B * 8 * N].view(torch.float32).view(B, 8, N).cpu())
            c = torch.stack([named[f"{fam}.{l}.attention_{h}.a.bias"].detach().cpu()[0] for fam in fams for h in range(4)])
            pre = uv[0][:, :, :, None] + uv[1][:, :, None, :] + c[None, :, None, None]
            out[(l, mod)] = (pre > 0, pre)
    return out


def aligned_oracle_backward(sd, inp_t, Lr, masks, upstream_fn, dev_signs):
    """Manual oracle forward, upstream gradients of `upstream_fn(outs)` by autograd on its (detached) outputs, manual
    backward with the device's LeakyReLU branches.  Returns (outs, gin, gw, dbg, stats); stats = flipped entries, total
    entries, and the largest |pre| (relative to max |pre| of its layer) among the flipped ones."""
    from oracle import unit_manual as MN
    outs, cache = MN.forward(sd, inp_t["A"], inp_t["M"], inp_t["dq"], inp_t["words"], inp_t["qlen"], Lr, masks=masks)
    flat = [outs[0], outs[1], outs[2]] + list(outs[3]) + list(outs[4]) + list(outs[5]) + list(outs[6])
    det = [o.detach().clone().requires_grad_(True) for o in flat]
    n = Lr
    d_outs = (det[0], det[1], det[2], det[3:3 + n], det[3 + n:3 + 2 * n], det[3 + 2 * n:3 + 3 * n], det[3 + 3 * n:3 + 4 * n])
    loss = upstream_fn(d_outs)
    gs = torch.autograd.grad(loss, det, allow_unused=True)
    gs = [g if g is not None else torch.zeros_like(o) for g, o in zip(gs, det)]
    flips = total = 0
    worst = 0.0
    pre_pos = {}
    for l in range(Lr):
        for mod in ("app", "mot"):
            cm = cache["layers"][l][mod]
            # the oracle's own logits: u, v are (B, 8, N); the a.bias of the 8 packed heads
            _, _, _, _, cc = MN.packed_gat(sd, mod, l)
            pre_o = cm["u"][:, :, :, None] + cm["v"][:, :, None, :] + cc[None, :, None, None]
            pos_d, _ = dev_signs[(l, mod)]
            diff = pos_d != (pre_o > 0)
            k = int(diff.sum())
            flips += k
            total += diff.numel()
            if k:
                worst = max(worst, float(pre_o[diff].abs().max()) / max(float(pre_o.abs().max()), 1e-30))
            pre_pos[(l, mod)] = pos_d
    gin, gw, dbg = MN.backward(sd, cache, inp_t["words"], inp_t["dq"], gs[0], gs[1], gs[2], gs[3:3 + n], gs[3 + n:3 + 2 * n],
                               gs[3 + 2 * n:3 + 3 * n], gs[3 + 3 * n:3 + 4 * n], pre_pos=pre_pos)
    return outs, gin, gw, dbg, {"flips": flips, "total": total, "worst_rel_pre": worst}


def assert_flips_at_kink(stats, what=""):
    """Every entry whose LeakyReLU branch differs from the oracle's lies at the kink, and they are rare."""
    print(f"  {what} LeakyReLU branch flips vs oracle: {stats['flips']} of {stats['total']} logits, "
          f"largest |pre| among them {stats['worst_rel_pre']:.1e} of the logit scale")
    assert stats["flips"] <= max(2, int(1e-4 * stats["total"])), stats
    assert stats["worst_rel_pre"] < 2e-4, stats


def grad_errors(m, t, gin, gw):
    """(worst input-gradient error, worst weight-gradient error, {a.bias name: |err| / sum|terms| scale unavailable})"""
    gerr = max(rel_err(t[k].grad.cpu(), gin[k]) for k in ("A", "M", "dq", "words"))
    werrs = {}
    for k, p in m.named_parameters():
        if k.endswith(".fc.bias"):
            continue                                   # analytically zero
        want = gw[k]
        werrs[k] = float((p.grad.cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-3)
    return gerr, werrs


def abias_scales(dbg, Lr):
    """{'<fam>.<l>.attention_<h>.a.bias': sum of |terms|} -- the scalar a.bias gradient is a sum of B*N*N signed terms
    that cancel to ~1e-3 of their magnitude, so its error is measured against the sum of the absolute terms."""
    out = {}
    for l in range(Lr):
        for mod, fams in (("app", ("acGCN", "appearance_GCN")), ("mot", ("mcGCN", "motion_GCN"))):
            for g, fam in enumerate(fams):
                for h in range(4):
                    out[f"{fam}.{l}.attention_{h}.a.bias"] = float(dbg["layers"][l][mod + ".dc_abs"][g][h])
    return out


def bias_scales(dbg, Lr, D):
    """{bias parameter name: per-element sum of |terms|} for the gradients that are column sums over rows (they cancel:
    a bias gradient can be 1e-3 of the sum of its terms' magnitudes)."""
    out = {}
    Fh = D // 4
    for l in range(Lr):
        dl = dbg["layers"][l]
        out[f"queryAttn.{l}.feat_enhance.bias"] = dl["feat_enhance.bias_abs"]
        for mod, fams, att, qp in (("app", ("acGCN", "appearance_GCN"), "attention_appearance", "queryPunish_appear"),
                                   ("mot", ("mcGCN", "motion_GCN"), "attention_motion", "queryPunish_motion")):
            ba = dl[mod + ".bias_abs"]
            out[f"{att}.{l}.project.0.bias"] = ba["project.0.bias"]
            out[f"{qp}.{l}.query_weight.bias"] = ba["query_weight.bias"]
            for g, fam in enumerate(fams):
                for h in range(4):
                    hh = g * 4 + h
                    out[f"{fam}.{l}.attention_{h}.W.bias"] = ba["W.bias"][hh * Fh:(hh + 1) * Fh]
    return out


def assert_grads(m, t, gin, gw, dbg, Lr, gtol, what="", scale_tol=None):
    """Input gradients and every weight-matrix gradient within `gtol` (max |err| / max |want| per tensor).  Gradients that
    are column sums over rows (the biases; the a.bias scalars) cancel heavily: they are held to 1e-5 of the sum of the
    absolute terms (what their arithmetic can deliver) and, as a sanity bound only, to 10 x gtol (bias vectors) / 40 x gtol
    (a.bias scalars) in the plain metric
    (a scalar a.bias gradient that cancels to 1e-3 of its terms shows 6e-3 there at bf16x3's 1e-5 operand error)."""
    gerr, werrs = grad_errors(m, t, gin, gw)
    if scale_tol is None:          # the terms themselves carry the operand mode's rounding: 1e-7 (fp32), 1e-5 (bf16x3)
        scale_tol = 5e-5 if "bf16x3" in what else 1e-5
    scales = abias_scales(dbg, Lr)
    named = dict(m.named_parameters())
    D = named["queryAttn.0.feat_enhance.weight"].shape[0]
    ab = {k: abs(float(named[k].grad.cpu()[0]) - float(gw[k][0])) / max(scales[k], 1e-30) for k in scales}
    bsc = bias_scales(dbg, Lr, D)
    bb = {k: float(((named[k].grad.cpu() - gw[k]).abs() / sc.clamp_min(1e-30)).max()) for k, sc in bsc.items()}
    special = set(scales) | set(bsc)
    plain = {k: v for k, v in werrs.items() if k not in special}
    worst = sorted(plain.items(), key=lambda kv: -kv[1])[:4]
    print(f"  {what} gin {gerr:.2e}  worst gw {[(k, f'{v:.2e}') for k, v in worst]}  a.bias/scale {max(ab.values()):.2e} "
          f"(plain {max(werrs[k] for k in scales):.2e})  bias/scale {max(bb.values()):.2e} (plain {max(werrs[k] for k in bsc):.2e})")
    assert gerr < gtol, gerr
    assert worst[0][1] < gtol, worst
    assert max(ab.values()) < scale_tol, sorted(ab.items(), key=lambda kv: -kv[1])[:3]
    assert max(bb.values()) < scale_tol, sorted(bb.items(), key=lambda kv: -kv[1])[:3]
    # plain-metric sanity bounds, per kind (round 6: one 40 x bound for both was wider than the vectors need): bias VECTORS
    # (max |err| / max |want| over D or 8 F entries: the largest entry does not cancel) 10 x gtol; the a.bias SCALARS -- one
    # number per head that is zero in exact arithmetic up to the kink terms (softmax shift invariance), cancelling to 1e-3 of
    # its terms -- 40 x gtol, their real gate being the Sigma|terms| one above.
    assert max(werrs[k] for k in bsc) < 10 * gtol, sorted(((k, werrs[k]) for k in bsc), key=lambda kv: -kv[1])[:3]
    assert max(werrs[k] for k in scales) < 40 * gtol, sorted(((k, werrs[k]) for k in scales), key=lambda kv: -kv[1])[:3]
