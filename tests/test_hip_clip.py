"""GPU: the appearance encoder (row f3: dropout -> tanh -> BiLSTM over the frames -> final states -> dropout,
include/dvgr_clip.h) through the drop-in module against
  (1) the golden vectors recorded from the reference's VisualAppearanceEncoder (eval mode; it runs torch.nn.LSTM),
  (2) the oracle (oracle/clip_torch.py) with the device's own dropout masks injected (training mode, both sites),
forward and backward (every LSTM parameter).  Tolerance: 1e-4 relative on the output, 5e-4 on gradients (fp32)."""
import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import clip as CL
from clip_helpers import CLIP_CASES, ClipCase
from helpers import rel_err
from oracle import clip_torch as CT

pytestmark = pytest.mark.gpu
TOL, GTOL = 1e-4, 5e-4


def build(case, dtype="fp32", train=False):
    m = dv.ClipEncoder(appearance_dim=case.V, module_dim=case.D, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)
    return m.cuda().train(train)


def device_masks(case, seed):
    lib = CL.load()
    out = {}
    for site, (nm, shape) in enumerate((("in", (case.B, case.N, case.T, case.V)), ("out", (case.B * case.N, case.D)))):
        keep = torch.empty(int(np.prod(shape)), device="cuda")
        assert lib.dvgr_clip_dbg_keep_mask(seed, site, keep.numel(), keep.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        out[nm] = keep.view(shape).cpu()
    return out


def oracle_run(case, train, masks):
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in case.sd_np.items()}
    out = CT.clip_forward(sd, torch.from_numpy(case.x.copy()), train=train, masks=masks)
    (out * torch.from_numpy(case.R)).sum().backward()
    return out.detach().numpy(), {k: v.grad.numpy() for k, v in sd.items()}


@pytest.mark.parametrize("name", [n for n in CLIP_CASES if n.endswith("eval")])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_clip_eval_matches_reference_golden(name, dtype):
    case = ClipCase(name)
    m = build(case, dtype)
    out = m(torch.from_numpy(case.x.copy()).cuda())
    (out * torch.from_numpy(case.R).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert out.shape == (case.B, case.N, case.D)
    assert rel_err(out.detach().cpu().numpy(), case.z["out"]) < TOL
    gtol = GTOL if dtype == "fp32" else 2e-3
    case.check_param_grads({k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, gtol, rel_err)


@pytest.mark.parametrize("name", ["g_clip_small_train", "g_clip_real_train"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_clip_training_matches_oracle_with_device_masks(name, dtype):
    case = ClipCase(name)
    m = build(case, dtype, train=True)
    m.forced_seed = 1234567
    out = m(torch.from_numpy(case.x.copy()).cuda())
    (out * torch.from_numpy(case.R).cuda()).sum().backward()
    torch.cuda.synchronize()
    masks = device_masks(case, m.forced_seed)
    assert abs(float(masks["in"].mean()) - 0.85) < 0.02 and abs(float(masks["out"].mean()) - 0.82) < 0.06
    want, wgrads = oracle_run(case, True, masks)
    assert rel_err(out.detach().cpu().numpy(), want) < TOL
    gtol = GTOL if dtype == "fp32" else 2e-3
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), wgrads[k]) < gtol, k


@pytest.mark.parametrize("shape", [(1, 1, 2, 32, 64), (3, 5, 7, 96, 128), (2, 20, 16, 256, 64), (5, 8, 3, 64, 192)])
def test_clip_other_sizes_match_oracle(shape):
    """(B, N, T, V, D) corners: one clip, odd row counts (the generic contraction path), T > S, wide hidden state."""
    B, N, T, V, D = shape

    class Case:
        pass
    case = Case()
    case.B, case.N, case.T, case.V, case.D = B, N, T, V, D
    from dualvgr_videoqa_amd import synth
    case.sd_np = synth.xavier_state_dict(CT.param_shapes(V, D), seed=99, bias_range=0.1)
    case.x = np.abs(synth.normal("clip_x", (B, N, T, V), 0.8, 99)).astype(np.float32)
    case.R = synth.normal("clip_R", (B, N, D), 1.0, 99)
    m = build(case, "fp32")
    out = m(torch.from_numpy(case.x.copy()).cuda())
    (out * torch.from_numpy(case.R).cuda()).sum().backward()
    want, wgrads = oracle_run(case, False, None)
    assert rel_err(out.detach().cpu().numpy(), want) < TOL
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), wgrads[k]) < GTOL, k


def test_clip_eval_is_deterministic_and_training_draws_new_masks():
    case = ClipCase("g_clip_small_train")
    m = build(case, "fp32")
    x = torch.from_numpy(case.x.copy()).cuda()
    a, b = m(x), m(x)
    assert torch.equal(a, b)
    m.train()
    c, d = m(x), m(x)
    assert not torch.equal(c, d)
    with pytest.raises(RuntimeError):
        m(x.cpu())
