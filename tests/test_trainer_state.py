"""CPU: checkpoint plumbing of the training step (reference train.py:360-365 saves `optimizer.state_dict()`, train.py:119
restores it).  `TrainStep.state_dict()` speaks torch.optim.Adam's format over `model.parameters()` order, so a reference
checkpoint's optimizer entry round-trips; ClipAdam's own flat state carries a layout tag and refuses foreign layouts."""
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import trainer


def _model():
    vocab = {"question_token_to_idx": {str(i): i for i in range(30)}, "answer_token_to_idx": {str(i): i for i in range(12)}}
    torch.manual_seed(0)
    return dv.DualVGR(vision_dim=32, module_dim=64, word_dim=32, vocab=vocab, num_of_nodes=8, graph_module="GAT",
                      graph_layers=1, unit_layers=2, dtype="fp32")


def _fill(ts, steps=3):
    g = torch.Generator().manual_seed(1)
    n = ts.unit.flat_params().numel()
    ts.opt._state["visual_input_unit"] = (torch.randn(n, generator=g), torch.rand(n, generator=g))
    for name, p in ts.other:
        ts.opt._state[name] = (torch.randn(p.numel(), generator=g), torch.rand(p.numel(), generator=g))
    ts.opt.step_count = steps


def test_state_dict_round_trips_through_torch_adam():
    m = _model()
    ts = dv.TrainStep(m, lr=3e-4)
    _fill(ts)
    sd = ts.state_dict()
    names = [n for n, _ in m.named_parameters()]
    assert sd["param_groups"][0]["params"] == list(range(len(names)))
    # torch's own optimizer accepts it (the format train.py:119 feeds to optimizer.load_state_dict) ...
    opt = torch.optim.Adam(m.parameters(), lr=1.0)
    opt.load_state_dict({k: v for k, v in sd.items() if k != "max_norm"})
    assert opt.param_groups[0]["lr"] == pytest.approx(3e-4)
    back = opt.state_dict()
    for i, (n, p) in enumerate(m.named_parameters()):
        assert tuple(back["state"][i]["exp_avg"].shape) == tuple(p.shape), n
        assert float(back["state"][i]["step"]) == 3.0
    # ... and what torch saves restores this package's flat moments exactly, through the parameter table (names, not offsets)
    ts2 = dv.TrainStep(_model(), lr=1.0)
    ts2.load_state_dict(back)
    assert ts2.opt.step_count == 3 and ts2.opt.lr == pytest.approx(3e-4)
    pre = ts._unit_prefix()
    used = torch.zeros(ts.unit.flat_params().numel(), dtype=torch.bool)
    for n, sh, o in ts.unit._table:
        k = 1
        for d in sh:
            k *= d
        used[o:o + k] = True
        assert pre + n in names
    for a, b in zip(ts.opt._state["visual_input_unit"], ts2.opt._state["visual_input_unit"]):
        assert torch.equal(a[used], b[used])
        assert float(b[~used].abs().sum()) == 0.0                  # padding between tensors carries no moments
    for name, _ in ts.other:
        for a, b in zip(ts.opt._state[name], ts2.opt._state[name]):
            assert torch.equal(a, b)


def test_load_rejects_other_models_and_mixed_steps():
    ts = dv.TrainStep(_model())
    _fill(ts)
    sd = ts.state_dict()
    bad = {"state": dict(sd["state"]), "param_groups": [dict(sd["param_groups"][0])]}
    bad["param_groups"][0]["params"] = bad["param_groups"][0]["params"][:-1]
    with pytest.raises(RuntimeError, match="parameters"):
        ts.load_state_dict(bad)
    bad2 = {"state": {k: dict(v) for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]}
    bad2["state"][0]["step"] = torch.tensor(9.0)
    with pytest.raises(RuntimeError, match="step counts"):
        ts.load_state_dict(bad2)
    bad3 = {"state": {k: dict(v) for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]}
    bad3["state"][0]["exp_avg"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="shape"):
        ts.load_state_dict(bad3)


def test_clipadam_flat_state_is_tagged_and_never_silently_reset():
    opt = trainer.ClipAdam()
    opt.set_layout("u", "layout-A")
    opt._state["u"] = (torch.ones(8), torch.ones(8))
    opt.step_count = 5
    sd = opt.state_dict()
    assert sd["layouts"] == {"u": "layout-A"}
    other = trainer.ClipAdam()
    other.set_layout("u", "layout-B")                              # same numel, different element order
    with pytest.raises(RuntimeError, match="layout"):
        other.load_state_dict(sd)
    legacy = {k: v for k, v in sd.items() if k not in ("format", "layouts")}
    with pytest.raises(RuntimeError, match="layout"):
        other.load_state_dict(legacy)                              # round-2 checkpoints carried no tag
    same = trainer.ClipAdam()
    same.set_layout("u", "layout-A")
    same.load_state_dict(sd)
    assert same.step_count == 5
    m, v = same._moments("u", torch.zeros(8))                      # kept, not zeroed
    assert float(m.sum()) == 8.0 and float(v.sum()) == 8.0
    with pytest.raises(RuntimeError, match="elements"):
        same._moments("u", torch.zeros(9))                         # a numel mismatch is an error, not a reset
