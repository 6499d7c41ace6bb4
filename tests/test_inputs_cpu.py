"""CPU: the host side of include/dvgr_inputs.h keeps the reference's parameter surface and refuses to run off the device."""
import pytest
import torch
import torch.nn as nn

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import inputs


def test_hip_linear_is_an_nn_linear_with_the_same_state_dict():
    lin = dv.HipLinear(2048, 768, dtype="bf16x3")
    assert isinstance(lin, nn.Linear)
    assert {k: tuple(v.shape) for k, v in lin.state_dict().items()} == {"weight": (768, 2048), "bias": (768,)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lin(torch.zeros(2, 2048))


def test_word_features_refuse_the_cpu():
    wf = inputs.WordFeatures()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        wf(torch.zeros(10, 8), torch.zeros(2, 3, dtype=torch.int64), False)
