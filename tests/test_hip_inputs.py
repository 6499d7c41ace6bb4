"""GPU: include/dvgr_inputs.h through its host side (dualvgr_videoqa_amd/inputs.py) against plain PyTorch fp32 on the same
device -- the word features tanh(dropout(embedding(q))) of model/Preprocessing.py:112-113 (dropout mask of the device's
counter hash injected into the torch expression) and the motion projection nn.Linear of model/models.py:46."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("shape", [(64, 15), (1, 1), (7, 40)])
def test_word_features_match_torch(train, shape):
    g = torch.Generator().manual_seed(5)
    vocab, wd = 97, 300
    table = (torch.rand(vocab, wd, generator=g) * 2 - 1).cuda().requires_grad_(True)
    tokens = torch.randint(0, vocab, shape, generator=g).cuda()
    tokens[0, 0] = tokens[-1, -1]                                   # at least one repeated token: the scatter must ADD
    wf = inputs.WordFeatures()
    wf.forced_seed = 1234 if train else None
    words = wf(table, tokens, train)
    up = torch.randn(*shape, wd, generator=g).cuda()
    words.backward(up)
    got_g = table.grad.clone()
    table.grad = None
    keep = inputs.words_keep_mask(1234, (*shape, wd), table.device) if train else torch.ones(*shape, wd, device="cuda")
    if train:
        frac = float(keep.mean())
        assert 0.80 < frac < 0.90 or keep.numel() < 2000           # p = 0.15
    want = torch.tanh(nn.functional.embedding(tokens, table) * keep / (0.85 if train else 1.0))
    want.backward(up)
    assert torch.equal(words, want) or float((words - want).abs().max()) < 2e-7
    assert float((got_g - table.grad).abs().max()) < 1e-5 * max(1.0, float(table.grad.abs().max()))
    wf.check_tokens()                                               # no bad token so far


def test_word_features_flag_a_token_outside_the_table():
    table = torch.zeros(10, 8).cuda()
    wf = inputs.WordFeatures()
    wf(table, torch.tensor([[1, 2, 10]]).cuda(), False)
    with pytest.raises(IndexError):
        wf.check_tokens()
    wf(table, torch.tensor([[1, 2, 9]]).cuda(), False)
    wf.check_tokens()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-6), ("bf16x3", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("shape", [(64, 20, 2048, 768), (2, 3, 64, 12), (1, 5, 300, 768)])
def test_linear_matches_torch(dtype, tol, shape):
    B, N, K, O = shape
    g = torch.Generator().manual_seed(11)
    lin = dv.HipLinear(K, O, dtype=dtype).cuda()
    ref = nn.Linear(K, O).cuda()
    ref.load_state_dict(lin.state_dict())                           # same keys, same shapes
    x = torch.randn(B, N, K, generator=g).cuda().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    up = torch.randn(B, N, O, generator=g).cuda()
    with torch.backends.cuda.sdp_kernel() if False else torch.no_grad():
        pass
    y = lin(x)
    yr = (xr.double() @ ref.weight.double().t() + ref.bias.double())
    y.backward(up)
    yr.backward(up.double())

    def rel(a, b):
        return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)

    assert rel(y, yr) < tol
    assert rel(lin.weight.grad, ref.weight.grad) < tol
    assert rel(lin.bias.grad, ref.bias.grad) < max(tol, 1e-5)
    assert rel(x.grad, xr.grad) < tol


def test_linear_without_input_gradient_and_bias():
    lin = dv.HipLinear(64, 16, bias=False, dtype="fp32").cuda()
    x = torch.randn(9, 64).cuda()
    y = lin(x)
    y.sum().backward()
    want = x.sum(0, keepdim=True).expand(16, 64)
    assert float((lin.weight.grad - want).abs().max()) < 1e-4
