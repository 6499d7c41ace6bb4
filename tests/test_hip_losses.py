"""GPU: the fused auxiliary-loss kernels (dvgr_scope_u_loss, csrc/dvgr_loss.hip) against the oracle's
restatement of utils.py:10-31 / train.py:148-154 evaluated in float64 on the CPU, and against the
`common` / `dep` values recorded from the imported reference in tests/golden/*.npz.
Tolerance: 1e-4 relative (max|got - want| / max|want|) per tensor, the gate of BASELINE.json."""
import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import losses, synth
from helpers import GoldenCase, rel_err
from oracle import unit_torch as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def oracle_terms_and_grads(lists, vis, alpha, beta, scale):
    """float64 CPU evaluation of the scope-U loss and its gradients."""
    t = [x.detach().cpu().double().requires_grad_(True) for x in lists]
    v = vis.detach().cpu().double().requires_grad_(True)
    ca, cm, aq, mq = t
    Lr, N = ca.shape[0], ca.shape[2]
    com, dep = O.aux_losses((v, None, None, list(ca), list(cm), list(aq), list(mq)), N)
    loss = v.sum() + alpha * com + beta * scale * dep
    loss.backward()
    return (float(v.detach().sum()), float(com.detach()), float(dep.detach())), [v.grad] + [x.grad for x in t]


def random_lists(Lr, B, N, D, seed, corr=0.0):
    out = []
    base = synth.normal("base", (Lr, B, 1, D), 1.0, seed)
    for nm in ("ca", "cm", "aq", "mq"):
        x = synth.normal(nm, (Lr, B, N, D), 0.5, seed)
        out.append(torch.from_numpy((x + corr * base).astype(np.float32)).cuda())
    vis = torch.from_numpy(synth.normal("vis", (B, N, D), 1.0, seed)).cuda()
    return out, vis


@pytest.mark.parametrize("Lr,B,N,D", [(3, 8, 20, 768), (2, 3, 8, 128), (1, 2, 33, 64), (2, 2, 64, 128), (1, 3, 16, 256),
                                      (1, 2, 1, 64), (1, 2, 2, 64), (4, 5, 24, 192),
                                      (2, 2, 64, 768), (1, 3, 48, 320), (1, 2, 40, 1024)])   # N > 32: the staged Gram pass, full and partial 64-column chunks
def test_fused_loss_against_float64_oracle(Lr, B, N, D):
    lists, vis = random_lists(Lr, B, N, D, seed=11 + N)
    alpha, beta, scale = 0.7, 3e-3, 2.0
    terms, ups = losses.fused_scope_u_loss((vis, None, None, *lists), alpha, beta, world_size=int(scale))
    torch.cuda.synchronize()
    want_t, want_g = oracle_terms_and_grads(lists, vis, alpha, beta, scale)
    got_t = terms.cpu().numpy()
    for k in range(3):
        assert abs(got_t[k] - want_t[k]) <= TOL * max(abs(want_t[k]), 1e-6), (k, got_t[k], want_t[k])
    assert ups[1] is None and ups[2] is None
    got_g = [ups[0]] + ups[3:]
    for name, g, w in zip(("vis", "com_app", "com_mot", "aq_fusion", "mq_fusion"), got_g, want_g):
        if float(w.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, name
        else:
            assert rel_err(g.cpu().numpy(), w.numpy()) < TOL, (name, rel_err(g.cpu().numpy(), w.numpy()))


def test_fused_loss_rows_with_a_large_shared_component():
    """Rows that share a large common component (what a deep GAT stack produces): the centring must happen on
    the rows, not on the Gram matrix, or the normalised Gram loses its digits."""
    lists, vis = random_lists(2, 4, 20, 256, seed=5, corr=30.0)
    terms, ups = losses.fused_scope_u_loss((vis, None, None, *lists), 1.0, 1e-4, 1)
    want_t, want_g = oracle_terms_and_grads(lists, vis, 1.0, 1e-4, 1.0)
    got_t = terms.cpu().numpy()
    for k in range(3):
        assert abs(got_t[k] - want_t[k]) <= 5e-4 * max(abs(want_t[k]), 1e-6), (k, got_t[k], want_t[k])
    for g, w in zip([ups[0]] + ups[3:], want_g):
        assert rel_err(g.cpu().numpy(), w.numpy()) < 5e-4


@pytest.mark.parametrize("name", ["g_real_n20_l3", "g_real_n8_l3", "g_small_eval"])
def test_fused_loss_on_unit_outputs_matches_golden_values(name):
    """Unit forward (HIP) -> fused loss: `common` and `dep` equal the values the reference produced."""
    case = GoldenCase(name)
    m = dv.DualVGRUnit_multiple(word_dim=case.Wd, module_dim=case.D, num_of_nodes=case.N, unit_layers=case.L)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)
    m = m.cuda().eval()
    t = case.torch_inputs(device="cuda")
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    terms, _ = losses.fused_scope_u_loss(raw, case.alpha, case.beta, 1)
    got = terms.cpu().numpy()
    assert abs(got[1] - float(case.z["common"])) <= 2e-4 * abs(float(case.z["common"])) + 1e-9
    assert abs(got[2] - float(case.z["dep"])) <= 2e-4 * abs(float(case.z["dep"])) + 1e-9


def test_autograd_node_end_to_end_equals_torch_losses():
    """ScopeULoss inside autograd: the parameter gradients of the unit equal those obtained with the plain
    PyTorch losses on the same device."""
    case = GoldenCase("g_real_n8_l3")
    grads = []
    for fused in (False, True):
        m = dv.DualVGRUnit_multiple(word_dim=case.Wd, module_dim=case.D, num_of_nodes=case.N, unit_layers=case.L,
                                    grad_mode="autograd")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)
        m = m.cuda().eval()
        t = case.torch_inputs(device="cuda", requires_grad=True)
        if fused:
            raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
            loss, _ = losses.ScopeULoss.apply(1.0, 1e-3, 1, raw[0], *raw[3:])
            loss = loss * 0.5
        else:
            outs = m(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
            loss = losses.scope_u_loss(outs, case.N, 1.0, 1e-3) * 0.5
        loss.backward()
        grads.append(([t[k].grad.cpu().numpy() for k in ("A", "M", "dq", "words")] +
                      [p.grad.cpu().numpy() for p in m.parameters()], float(loss)))
    assert abs(grads[0][1] - grads[1][1]) <= 1e-5 * abs(grads[0][1])
    scale = max(float(np.abs(a).max()) for a in grads[0][0])
    for a, b in zip(grads[0][0], grads[1][0]):
        if a.size == 1:       # scalar biases: sums that cancel to ~0 (fc.bias is analytically 0), absolute check
            assert abs(float(a.ravel()[0]) - float(b.ravel()[0])) < 1e-5 * scale
        elif np.abs(a).max() > 1e-12:
            assert rel_err(b, a) < 5e-4


def test_aux_terms_alone_without_vis():
    """vis = None: alpha * common + beta * dependence only (the form train.py:148-154 adds to the cross-entropy)."""
    lists, vis = random_lists(2, 3, 20, 128, seed=9)
    ls = [t.clone().requires_grad_(True) for t in lists]
    aux, terms = losses.ScopeULoss.apply(0.5, 2e-3, 1, None, *ls)
    (3.0 * aux).backward()
    want_t, want_g = oracle_terms_and_grads(lists, torch.zeros_like(vis), 0.5, 2e-3, 1.0)
    assert float(terms[0]) == 0.0
    assert abs(float(aux) - (0.5 * want_t[1] + 2e-3 * want_t[2])) <= TOL * abs(0.5 * want_t[1] + 2e-3 * want_t[2])
    for t, w in zip(ls, want_g[1:]):
        assert rel_err(t.grad.cpu().numpy(), 3.0 * w.numpy()) < TOL


@pytest.mark.parametrize("train", [False, True])
def test_loss_evaluated_by_the_forward_equals_the_separate_call(train):
    """dvgr_opts.loss (kernels enqueued by the forward, on the side stream) == dvgr_scope_u_loss afterwards."""
    case = GoldenCase("g_real_n20_l3")
    m = dv.DualVGRUnit_multiple(word_dim=case.Wd, module_dim=case.D, num_of_nodes=case.N, unit_layers=case.L)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)
    m = m.cuda().train(train)
    m.forced_seed = 77
    m.loss_in_forward = (0.7, 3e-3, 2.0)
    t = case.torch_inputs(device="cuda")
    raw = m.forward_raw(t["A"], t["M"], t["dq"], t["words"], t["qlen"])
    terms, ups = m.last_loss
    terms2, ups2 = losses.fused_scope_u_loss(raw, 0.7, 3e-3, 2)
    torch.cuda.synchronize()
    assert torch.allclose(terms, terms2, rtol=1e-6, atol=0)
    for a, b in zip(ups, ups2):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
