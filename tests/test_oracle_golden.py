"""CPU: the oracle (oracle/unit_torch.py) against the golden vectors recorded from the imported
reference (tests/golden/make_golden.py).  Tolerances: fp32 reassociation only (1e-5 relative on
outputs, 2e-4 on gradients of the chained fp32 graph)."""
import numpy as np
import pytest
import torch

from helpers import REAL_CASES, SMALL_CASES, GoldenCase, rel_err, scalar_loss
from oracle import unit_torch as O

OUT_TOL = 1e-5
GRAD_TOL = 2e-4


def _run(case: GoldenCase, naive: bool):
    sd = {k: v.requires_grad_(True) for k, v in case.torch_state().items()}
    t = case.torch_inputs(requires_grad=True)
    outs = O.unit_forward(sd, t["A"], t["M"], t["dq"], t["words"], t["qlen"], case.L, case.G,
                          masks=case.masks(), naive=naive)
    R = {k: torch.from_numpy(v) for k, v in case.upstream().items()}
    loss, com, dep = scalar_loss(outs, R, case.N, case.alpha, case.beta)
    loss.backward()
    return sd, t, outs, loss, com, dep


@pytest.mark.parametrize("naive", [True, False], ids=["naive", "factored"])
@pytest.mark.parametrize("name", SMALL_CASES)
def test_small_cases_full(name, naive):
    case = GoldenCase(name)
    z = case.z
    sd, t, outs, loss, com, dep = _run(case, naive)
    vis, aqe, mqe, ca, cm, aq, mq = outs
    assert rel_err(vis.detach(), z["out/vis"]) < OUT_TOL
    assert rel_err(aqe.detach(), z["out/aq_embed"]) < OUT_TOL
    assert rel_err(mqe.detach(), z["out/mq_embed"]) < OUT_TOL
    assert len(aq) == case.L * case.G
    for l in range(len(aq)):
        assert rel_err(ca[l].detach(), z[f"out/com_app{l}"]) < OUT_TOL
        assert rel_err(cm[l].detach(), z[f"out/com_motion{l}"]) < OUT_TOL
        assert rel_err(aq[l].detach(), z[f"out/aq_fusion{l}"]) < OUT_TOL
        assert rel_err(mq[l].detach(), z[f"out/mq_fusion{l}"]) < OUT_TOL
    assert abs(float(loss) - float(z["loss"])) < 1e-4 * max(1.0, abs(float(z["loss"])))
    if case.alpha != 0.0:   # alpha == 0 marks a case whose common_loss is ill-conditioned (near-identical rows)
        assert abs(float(com) - float(z["common"])) < 1e-5 * max(1.0, abs(float(z["common"])))
    assert abs(float(dep) - float(z["dep"])) < 1e-4 * max(1.0, abs(float(z["dep"])))
    for k in ("A", "M", "dq", "words"):
        assert rel_err(t[k].grad, z["gin/" + k]) < GRAD_TOL, k
    big = set(z["gwbig_names"].tolist()) if "gwbig_names" in z.files else set()
    for k, p in sd.items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if k.endswith(".fc.bias"):
            # analytically zero (softmax over T is shift invariant): only rounding noise remains
            assert float(g.abs().max()) < 1e-4 and float(np.abs(z["gw/" + k]).max()) < 1e-4
        elif k in big:
            i = sorted(big).index(k)
            assert abs(float(g.double().sum()) - z["gwbig_sum"][i]) < GRAD_TOL * max(z["gwbig_abs"][i], 1e-2), k
            assert np.allclose(g.flatten()[:64].numpy(), z["gwbig_head"][i], rtol=0, atol=GRAD_TOL * float(np.abs(z["gwbig_head"][i]).max() + 1e-2)), k
        else:
            want = z["gw/" + k]
            scale = max(float(np.abs(want).max()), 1e-3)  # fc.bias grad is analytically 0 (softmax shift)
            assert float(np.abs(g.numpy() - want).max()) < GRAD_TOL * scale, k


@pytest.mark.parametrize("name", REAL_CASES)
def test_real_dims(name):
    case = GoldenCase(name)
    z = case.z
    sd, t, outs, loss, com, dep = _run(case, naive=False)
    vis, aqe, mqe, ca, cm, aq, mq = outs
    last = len(aq) - 1
    assert rel_err(vis.detach(), z["out/vis"]) < OUT_TOL
    assert rel_err(mqe.detach(), z["out/mq_embed"]) < OUT_TOL
    assert rel_err(ca[last].detach(), z[f"out/com_app{last}"]) < OUT_TOL
    assert rel_err(aq[last].detach(), z[f"out/aq_fusion{last}"]) < OUT_TOL
    for k in ("A", "M", "dq", "words"):
        assert rel_err(t[k].grad, z["gin/" + k]) < GRAD_TOL, k
    names = z["gw_names"].tolist()
    for i, k in enumerate(names):
        g = sd[k].grad
        if k.endswith(".fc.bias"):
            assert float(g.abs().max()) < 1e-3
            continue
        assert abs(float(g.double().sum()) - z["gw_sum"][i]) < GRAD_TOL * max(z["gw_abs"][i], 1e-2), k
        head = np.resize(g.flatten()[:8].numpy(), 8)
        assert np.allclose(head, z["gw_head"][i], rtol=0, atol=GRAD_TOL * float(np.abs(z["gw_head"][i]).max() + 1e-2)), k


def test_naive_real_dims_forward():
    """The shape-faithful (B,N,N,2F) form that bench.py times as the CPU baseline."""
    case = GoldenCase("g_real_n20_l1")
    with torch.no_grad():
        t = case.torch_inputs()
        outs = O.unit_forward(case.torch_state(), t["A"], t["M"], t["dq"], t["words"], t["qlen"], case.L, case.G, naive=True)
    assert rel_err(outs[0], case.z["out/vis"]) < OUT_TOL


def test_param_inventory_matches_survey():
    """unit parameters = 0.985 M (MFB) + 4.605 M per layer at D=768, Wd=300."""
    n1 = sum(int(np.prod(s)) for s in O.param_shapes(300, 768, 1).values())
    n3 = sum(int(np.prod(s)) for s in O.param_shapes(300, 768, 3).values())
    assert n1 == 5_590_789 or abs(n1 - 5.59e6) < 1e4
    assert abs((n3 - n1) / 2 - 4.605e6) < 5e3
