"""CPU: the appearance-encoder oracle (oracle/clip_torch.py, the LSTM written out step by step) against the golden
vectors recorded from the reference's VisualAppearanceEncoder, which runs torch.nn.LSTM (tests/golden/g_clip_*.npz)."""
import pytest
import torch

from clip_helpers import CLIP_CASES, QUESTION_CASES, ClipCase, QuestionCase
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


@pytest.mark.parametrize("name", QUESTION_CASES)
def test_question_oracle_matches_reference(name):
    """Packing semantics restated as per-step activity masks against the reference's pack_padded_sequence path."""
    case = QuestionCase(name)
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in case.sd_np.items()}
    qe, words, seq = CT.question_forward(sd, torch.from_numpy(case.q), torch.from_numpy(case.qlen), train=case.train,
                                         masks=case.masks())
    R = {k: torch.from_numpy(v) for k, v in case.R.items()}
    ((qe * R["qe"]).sum() + (words * R["words"]).sum() + (seq * R["seq"]).sum()).backward()
    assert rel_err(qe.detach().numpy(), case.z["qe"]) < 1e-5
    assert rel_err(words.detach().numpy(), case.z["words"]) < 1e-6
    assert rel_err(seq.detach().numpy(), case.z["seq"]) < 1e-5
    for t in range(case.T):                                   # zero past each question's length
        assert float(seq.detach()[torch.from_numpy(case.qlen) <= t, t].abs().sum()) == 0.0
    for k, v in sd.items():
        assert rel_err(v.grad.numpy(), case.z["g_" + k]) < 2e-4, k
