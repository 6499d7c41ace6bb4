"""CPU: the hand-derived backward (oracle/unit_manual.py -- the algebra the HIP kernels implement)
against autograd through the op-for-op restatement (oracle/unit_torch.py), in float64."""
import numpy as np
import pytest
import torch

from helpers import GoldenCase, scalar_loss
from oracle import unit_manual as MN
from oracle import unit_torch as O


@pytest.mark.parametrize("name", ["g_small_eval", "g_small_train", "g_quirk_n1"])
def test_manual_backward_matches_autograd(name):
    case = GoldenCase(name)
    f64 = torch.float64
    sd = {k: v.requires_grad_(True) for k, v in case.torch_state(dtype=f64).items()}
    t = case.torch_inputs(dtype=f64, requires_grad=True)
    masks = case.masks()
    outs = O.unit_forward(sd, t["A"], t["M"], t["dq"], t["words"], t["qlen"], case.L, 1, masks=masks, naive=False)
    R = {k: torch.from_numpy(v).to(f64) for k, v in case.upstream().items()}
    loss, _, _ = scalar_loss(outs, R, case.N, case.alpha, case.beta)
    flat_outs = [outs[0], outs[1], outs[2]] + outs[3] + outs[4] + outs[5] + outs[6]
    loss.backward()
    # the DIRECT upstream gradients an autograd.Function.backward would receive: differentiate the
    # loss w.r.t. detached copies of the outputs (not through their downstream uses inside the unit)
    L_ = case.L
    det = [o.detach().clone().requires_grad_(True) for o in flat_outs]
    douts = (det[0], det[1], det[2], det[3:3 + L_], det[3 + L_:3 + 2 * L_], det[3 + 2 * L_:3 + 3 * L_], det[3 + 3 * L_:])
    loss2, _, _ = scalar_loss(douts, R, case.N, case.alpha, case.beta)
    gouts = torch.autograd.grad(loss2, det)

    sd0 = {k: v.detach() for k, v in sd.items()}
    t0 = {k: v.detach() for k, v in t.items()}
    mouts, cache = MN.forward(sd0, t0["A"], t0["M"], t0["dq"], t0["words"], t0["qlen"], case.L, masks=masks)
    mflat = [mouts[0], mouts[1], mouts[2]] + mouts[3] + mouts[4] + mouts[5] + mouts[6]
    for got, want in zip(mflat, flat_outs):
        assert torch.allclose(got, want.detach(), rtol=1e-10, atol=1e-12)
    L = case.L
    g = list(gouts)
    gin, gw, _ = MN.backward(sd0, cache, t0["words"], t0["dq"], g[0], g[1], g[2],
                             g[3:3 + L], g[3 + L:3 + 2 * L], g[3 + 2 * L:3 + 3 * L], g[3 + 3 * L:3 + 4 * L])
    for k in ("A", "M", "dq", "words"):
        assert torch.allclose(gin[k], t[k].grad, rtol=1e-9, atol=1e-11), k
    for k, p in sd.items():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(gw[k], want, rtol=1e-8, atol=1e-10), k


def test_aligned_backward_helper_equals_autograd_and_counts_flips():
    """helpers.aligned_oracle_backward (what the GPU parity tests compare gradients with): fed with the oracle's OWN logit
    signs it reports zero flips and reproduces autograd; with one sign forced to the other side it reports that flip and
    the gradients move (the LeakyReLU derivative really is discontinuous there)."""
    from helpers import aligned_oracle_backward
    case = GoldenCase("g_small_eval")
    f64 = torch.float64
    sd = case.torch_state(dtype=f64)
    t = case.torch_inputs(dtype=f64)
    R = {k: torch.from_numpy(v).to(f64) for k, v in case.upstream().items()}
    _, cache = MN.forward(sd, t["A"], t["M"], t["dq"], t["words"], t["qlen"], case.L)
    signs = {}
    for l in range(case.L):
        for mod in ("app", "mot"):
            cm = cache["layers"][l][mod]
            cc = MN.packed_gat(sd, mod, l)[4]
            pre = cm["u"][:, :, :, None] + cm["v"][:, :, None, :] + cc[None, :, None, None]
            signs[(l, mod)] = (pre > 0, pre)
    up = lambda o: scalar_loss(o, R, case.N, case.alpha, case.beta)[0]
    _, gin, gw, dbg, stats = aligned_oracle_backward(sd, t, case.L, None, up, signs)
    assert stats["flips"] == 0 and stats["total"] == 2 * case.L * case.B * 8 * case.N * case.N
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tg = case.torch_inputs(dtype=f64, requires_grad=True)
    outs = O.unit_forward(sdg, tg["A"], tg["M"], tg["dq"], tg["words"], tg["qlen"], case.L, 1, naive=False)
    scalar_loss(outs, R, case.N, case.alpha, case.beta)[0].backward()
    for k in ("A", "M", "dq", "words"):
        assert torch.allclose(gin[k], tg[k].grad, rtol=1e-9, atol=1e-11), k
    assert all(len(dbg["layers"][l]["app.dc_abs"]) == 2 for l in range(case.L))
    pos, pre = signs[(0, "app")]
    pos = pos.clone()
    pos[0, 0, 0, 0] = ~pos[0, 0, 0, 0]
    signs[(0, "app")] = (pos, pre)
    _, gin2, _, _, stats2 = aligned_oracle_backward(sd, t, case.L, None, up, signs)
    assert stats2["flips"] == 1 and not torch.allclose(gin2["A"], gin["A"], rtol=1e-9, atol=1e-11)
