"""CPU: the answer-head oracle (oracle/head_torch.py) against the golden vectors recorded from the reference's
ContextSelfAttn / SimpleOutputUnitOpenEnded (tests/golden/g_head_*.npz, make_golden_head.py)."""
import numpy as np
import pytest
import torch

from head_helpers import HEAD_CASES, HeadCase
from helpers import rel_err
from oracle import head_torch as H


@pytest.mark.parametrize("name", HEAD_CASES)
def test_head_oracle_matches_reference(name):
    case = HeadCase(name)
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in case.sd_np.items()}
    vis = torch.from_numpy(case.vis.copy()).requires_grad_(True)
    qe = torch.from_numpy(case.qe.copy()).requires_grad_(True)
    running = {"mean": torch.from_numpy(case.run_mean.copy()), "var": torch.from_numpy(case.run_var.copy())}
    logits, aux = H.head_forward(sd, vis, qe, train=case.train, masks=case.masks(), running=running)
    (logits * torch.from_numpy(case.R)).sum().backward()
    assert rel_err(logits.detach().numpy(), case.z["logits"]) < 1e-5
    assert rel_err(aux["v_distill"].detach().numpy(), case.z["v_distill"]) < 1e-5
    assert rel_err(vis.grad.numpy(), case.z["g_vis"]) < 2e-4
    assert rel_err(qe.grad.numpy(), case.z["g_qe"]) < 2e-4
    if case.train:
        assert rel_err(aux["new_running_mean"].numpy(), case.z["new_run_mean"]) < 1e-5
        assert rel_err(aux["new_running_var"].numpy(), case.z["new_run_var"]) < 1e-5
    case.check_param_grads({k: v.grad.numpy() for k, v in sd.items()}, tol=2e-4)
