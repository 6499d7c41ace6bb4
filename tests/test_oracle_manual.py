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
