"""CPU: the appearance-encoder oracle (oracle/clip_torch.py, the LSTM written out step by step) against the golden
vectors recorded from the reference's VisualAppearanceEncoder, which runs torch.nn.LSTM (tests/golden/g_clip_*.npz)."""
import pytest
import torch

from clip_helpers import CLIP_CASES, ClipCase
from helpers import rel_err
from oracle import clip_torch as CT


@pytest.mark.parametrize("name", CLIP_CASES)
def test_clip_oracle_matches_reference(name):
    case = ClipCase(name)
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in case.sd_np.items()}
    out = CT.clip_forward(sd, torch.from_numpy(case.x.copy()), train=case.train, masks=case.masks())
    (out * torch.from_numpy(case.R)).sum().backward()
    assert out.shape == (case.B, case.N, case.D)
    assert rel_err(out.detach().numpy(), case.z["out"]) < 1e-5
    case.check_param_grads({k: v.grad.numpy() for k, v in sd.items()}, 2e-4, rel_err)
