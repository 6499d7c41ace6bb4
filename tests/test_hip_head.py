"""GPU: the answer head (row f2: clip read-out + classifier, include/dvgr_head.h) through the drop-in module against
  (1) the golden vectors recorded from the reference's ContextSelfAttn / SimpleOutputUnitOpenEnded (eval mode),
  (2) the oracle (oracle/head_torch.py) with the device's own dropout masks injected (training mode: all three sites,
      batch statistics, running-statistics update), forward and backward.
Tolerance: 1e-4 relative on the logits (BASELINE.json's gate), 5e-4 on gradients."""
import ctypes as C

import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import head as HD
from head_helpers import HEAD_CASES, HeadCase
from helpers import rel_err
from oracle import head_torch as H

pytestmark = pytest.mark.gpu
TOL, GTOL = 1e-4, 5e-4


def build(case, dtype="fp32", train=False):
    m = dv.AnswerHead(module_dim=case.D, num_answers=case.C, dtype=dtype)
    sd = {k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}
    sd[H.PREFIX_O + "classifier.3.running_mean"] = torch.from_numpy(case.run_mean.copy())
    sd[H.PREFIX_O + "classifier.3.running_var"] = torch.from_numpy(case.run_var.copy())
    sd[H.PREFIX_O + "classifier.3.num_batches_tracked"] = torch.tensor(0)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train(train)
    return m


def device_masks(case, seed):
    lib = HD.load()
    out = {}
    for site, (nm, shape) in enumerate((("agg", (case.B, case.N, case.D)), ("cat", (case.B, 2 * case.D)), ("bn", (case.B, case.D)))):
        keep = torch.empty(int(np.prod(shape)), device="cuda")
        assert lib.dvgr_head_dbg_keep_mask(seed, site, keep.numel(), keep.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        out[nm] = keep.view(shape).cpu()
    return out


@pytest.mark.parametrize("name", [n for n in HEAD_CASES if n.endswith("eval")])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_head_eval_matches_reference_golden(name, dtype):
    case = HeadCase(name)
    m = build(case, dtype)
    vis = torch.from_numpy(case.vis.copy()).cuda().requires_grad_(True)
    qe = torch.from_numpy(case.qe.copy()).cuda().requires_grad_(True)
    logits = m(qe, vis)
    (logits * torch.from_numpy(case.R).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert logits.shape == (case.B, case.C)
    assert rel_err(logits.detach().cpu().numpy(), case.z["logits"]) < TOL
    gtol = GTOL if dtype == "fp32" else 2e-3
    assert rel_err(vis.grad.cpu().numpy(), case.z["g_vis"]) < gtol
    assert rel_err(qe.grad.cpu().numpy(), case.z["g_qe"]) < gtol
    case.check_param_grads({k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, tol=gtol)


@pytest.mark.parametrize("name", [n for n in HEAD_CASES if n.endswith("train")])
def test_head_training_mode_against_oracle_with_device_masks(name):
    case = HeadCase(name)
    seed = 20260301
    m = build(case, "fp32", train=True)
    m.forced_seed = seed
    vis = torch.from_numpy(case.vis.copy()).cuda().requires_grad_(True)
    qe = torch.from_numpy(case.qe.copy()).cuda().requires_grad_(True)
    logits = m(qe, vis)
    (logits * torch.from_numpy(case.R).cuda()).sum().backward()
    torch.cuda.synchronize()
    masks = device_masks(case, seed)
    for mk in masks.values():                                   # the RNG drops about 15 %
        assert 0.05 < 1.0 - float(mk.mean()) < 0.3 or mk.numel() < 200
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in case.sd_np.items()}
    ov = torch.from_numpy(case.vis.copy()).requires_grad_(True)
    oq = torch.from_numpy(case.qe.copy()).requires_grad_(True)
    running = {"mean": torch.from_numpy(case.run_mean.copy()), "var": torch.from_numpy(case.run_var.copy())}
    ol, aux = H.head_forward(sd, ov, oq, train=True, masks=masks, running=running)
    (ol * torch.from_numpy(case.R)).sum().backward()
    assert rel_err(logits.detach().cpu().numpy(), ol.detach().numpy()) < TOL
    assert rel_err(vis.grad.cpu().numpy(), ov.grad.numpy()) < GTOL
    assert rel_err(qe.grad.cpu().numpy(), oq.grad.numpy()) < GTOL
    bn = m.output_unit.classifier[3]
    assert rel_err(bn.running_mean.cpu().numpy(), aux["new_running_mean"].numpy()) < 1e-5
    assert rel_err(bn.running_var.cpu().numpy(), aux["new_running_var"].numpy()) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    for k, p in m.named_parameters():
        w = sd[k].grad.numpy()
        if k.endswith("attn.bias"):                              # analytically zero (shift invariance of the softmax)
            continue
        assert rel_err(p.grad.cpu().numpy(), w) < GTOL, k


def test_head_num_answers_not_a_multiple_of_four_and_state_dict_names():
    m = dv.AnswerHead(module_dim=64, num_answers=10).cuda().eval()
    assert set(m.state_dict().keys()) == set(H.param_shapes(64, 10).keys()) | {
        H.PREFIX_O + "classifier.3.running_mean", H.PREFIX_O + "classifier.3.running_var",
        H.PREFIX_O + "classifier.3.num_batches_tracked"}
    vis = torch.randn(3, 7, 64, device="cuda", requires_grad=True)
    qe = torch.randn(3, 64, device="cuda", requires_grad=True)
    logits = m(qe, vis)
    assert logits.shape == (3, 10)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    running = {"mean": sd[H.PREFIX_O + "classifier.3.running_mean"], "var": sd[H.PREFIX_O + "classifier.3.running_var"]}
    want, _ = H.head_forward(sd, vis.detach().cpu(), qe.detach().cpu(), train=False, running=running)
    assert rel_err(logits.detach().cpu().numpy(), want.numpy()) < TOL
    logits.sum().backward()
    assert m.output_unit.classifier[5].weight.grad.shape == (10, 64)


@pytest.mark.parametrize("Cn", [10, 1000])
def test_fused_cross_entropy_matches_torch(Cn):
    m = dv.AnswerHead(module_dim=64, num_answers=Cn).cuda().eval()
    vis = torch.randn(5, 7, 64, device="cuda")
    qe = torch.randn(5, 64, device="cuda")
    logits = m(qe, vis)
    target = torch.randint(0, Cn, (5,), device="cuda")
    loss, g = m.cross_entropy(logits, target, scale=0.5)
    ref_logits = logits.detach().clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(ref_logits, target)
    (0.5 * want).backward()
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    assert rel_err(g.cpu().numpy(), ref_logits.grad.cpu().numpy()) < 1e-5
