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
from clip_helpers import CLIP_CASES, QUESTION_CASES, ClipCase, QuestionCase
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


# ---- question encoder: two packed BiLSTMs over the words (model/Preprocessing.py:89-127, :7-45) ---------------------------
@pytest.mark.parametrize("name", [n for n in QUESTION_CASES if n.endswith("eval")])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_question_encoder_eval_matches_reference_golden(name, dtype):
    case = QuestionCase(name)
    m = CL.QuestionEncoder(case.VQ, case.Wd, case.D, dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in case.sd_np.items()}, strict=True)
    m = m.cuda().eval()
    qe, words, seq = m(torch.from_numpy(case.q).cuda(), torch.from_numpy(case.qlen).cuda())
    R = {k: torch.from_numpy(v).cuda() for k, v in case.R.items()}
    ((qe * R["qe"]).sum() + (words * R["words"]).sum() + (seq * R["seq"]).sum()).backward()
    torch.cuda.synchronize()
    assert rel_err(qe.detach().cpu().numpy(), case.z["qe"]) < TOL
    assert rel_err(words.detach().cpu().numpy(), case.z["words"]) < 1e-6
    assert rel_err(seq.detach().cpu().numpy(), case.z["seq"]) < TOL
    qlen = torch.from_numpy(case.qlen)
    for t in range(case.T):
        assert float(seq.detach().cpu()[qlen <= t, t].abs().sum()) == 0.0
    gtol = GTOL if dtype == "fp32" else 2e-3
    for k, p in m.named_parameters():                 # includes the embedding table (through g_x of both LSTMs)
        assert rel_err(p.grad.cpu().numpy(), case.z["g_" + k]) < gtol, k


@pytest.mark.parametrize("shape", [(4, 6, 20, 32), (64, 15, 300, 384), (7, 1, 8, 32), (130, 9, 64, 64)])
@pytest.mark.parametrize("train", [False, True])
def test_packed_bilstm_matches_oracle(shape, train):
    """(S, T, V, H): BiLSTM.run with lengths, both outputs and the input gradient; training mode adds the 0.18 dropout on
    the final states with the device's own mask injected into the oracle."""
    from dualvgr_videoqa_amd import synth
    S, T, V, H = shape
    sd_np = synth.xavier_state_dict(CT.lstm_shapes("", V, H), seed=31, bias_range=0.1)
    x_np = synth.normal("bl_x", (S, T, V), 0.6, 31)
    lens_np = (1 + (synth.uniform("bl_len", (S,), 0.0, 1.0, 31) * T).astype(np.int64)).clip(1, T)
    lens_np[0] = T
    Rf, Rs = synth.normal("bl_Rf", (S, 2 * H), 1.0, 31), synth.normal("bl_Rs", (S, T, 2 * H), 1.0, 31)
    m = CL.BiLSTM(V, H, dtype="fp32")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    m = m.cuda().train(train)
    m.forced_seed = 424242
    x = torch.from_numpy(x_np.copy()).cuda().requires_grad_(True)
    final, seq = m.run(x, torch.from_numpy(lens_np).cuda(), CL.LSTM_DROP_FINAL, want_final=True, want_seq=True)
    ((final * torch.from_numpy(Rf).cuda()).sum() + (seq * torch.from_numpy(Rs).cuda()).sum()).backward()
    torch.cuda.synchronize()
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
    xo = torch.from_numpy(x_np.copy()).requires_grad_(True)
    of, os_ = CT.packed_bilstm(sd, "", xo, torch.from_numpy(lens_np))
    if train:
        keep = torch.empty(S * 2 * H, device="cuda")
        assert CL.load().dvgr_clip_dbg_keep_mask(m.forced_seed, 1, keep.numel(), keep.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream) == 0
        of = of * keep.view(S, 2 * H).cpu() / (1.0 - CT.DROP_OUT)
    ((of * torch.from_numpy(Rf)).sum() + (os_ * torch.from_numpy(Rs)).sum()).backward()
    assert rel_err(final.detach().cpu().numpy(), of.detach().numpy()) < TOL
    assert rel_err(seq.detach().cpu().numpy(), os_.detach().numpy()) < TOL
    assert rel_err(x.grad.cpu().numpy(), xo.grad.numpy()) < GTOL
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), sd[k].grad.numpy()) < GTOL, k


@pytest.mark.parametrize("shape", [(256, 5, 32, 64), (512, 4, 32, 384), (256, 3, 64, 128), (1280, 3, 32, 384)])
@pytest.mark.parametrize("with_lens", [False, True])
def test_bilstm_bf16x3_at_bench_scale_matches_oracle(shape, with_lens):
    """(S, T, V, H) at the row counts the bench runs (the last shape: the appearance encoder's 1280 sequences x 384 units) in the
    bench's dtype: both outputs, the input gradient and every parameter gradient against the packed-LSTM oracle, with and
    without lengths.  (These are also the shapes the one-launch recurrence experiment of round 3 takes -- S a multiple of 256,
    H in {64, 128, 256, 384}: profiles/patches/r03_one_launch_recurrence.patch was held to this test.)"""
    from dualvgr_videoqa_amd import synth
    S, T, V, H = shape
    sd_np = synth.xavier_state_dict(CT.lstm_shapes("", V, H), seed=47, bias_range=0.1)
    x_np = synth.normal("rl_x", (S, T, V), 0.6, 47)
    lens_np = (1 + (synth.uniform("rl_len", (S,), 0.0, 1.0, 47) * T).astype(np.int64)).clip(1, T)
    lens_np[0] = T
    if not with_lens:
        lens_np[:] = T
    Rf, Rs = synth.normal("rl_Rf", (S, 2 * H), 1.0, 47), synth.normal("rl_Rs", (S, T, 2 * H), 1.0, 47)
    m = CL.BiLSTM(V, H, dtype="bf16x3")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(x_np.copy()).cuda().requires_grad_(True)
    lens = torch.from_numpy(lens_np).cuda() if with_lens else None
    final, seq = m.run(x, lens, 0, want_final=True, want_seq=True)
    ((final * torch.from_numpy(Rf).cuda()).sum() + (seq * torch.from_numpy(Rs).cuda()).sum()).backward()
    torch.cuda.synchronize()
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
    xo = torch.from_numpy(x_np.copy()).requires_grad_(True)
    of, os_ = CT.packed_bilstm(sd, "", xo, torch.from_numpy(lens_np))
    ((of * torch.from_numpy(Rf)).sum() + (os_ * torch.from_numpy(Rs)).sum()).backward()
    assert rel_err(final.detach().cpu().numpy(), of.detach().numpy()) < TOL
    assert rel_err(seq.detach().cpu().numpy(), os_.detach().numpy()) < TOL
    assert rel_err(x.grad.cpu().numpy(), xo.grad.numpy()) < 2e-3
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), sd[k].grad.numpy()) < 2e-3, k


def test_bilstm_single_output_variants():
    """Only the final states / only the per-step states requested: same values, gradients flow through the one asked for."""
    from dualvgr_videoqa_amd import synth
    S, T, V, H = 6, 5, 16, 32
    m = CL.BiLSTM(V, H).cuda().eval()
    x = torch.from_numpy(synth.normal("bl_x", (S, T, V), 0.6, 5)).cuda()
    lens = torch.tensor([5, 1, 3, 2, 5, 4], device="cuda")
    f_both, s_both = m.run(x, lens, 0, True, True)
    f_only, none1 = m.run(x, lens, 0, True, False)
    none2, s_only = m.run(x, lens, 0, False, True)
    assert none1 is None and none2 is None
    assert torch.equal(f_both, f_only) and torch.equal(s_both, s_only)
    s_only.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_packed_bilstm_random_shape_sweep():
    """20 seeded random (S, T, V, H, lengths, which outputs) draws against the oracle: odd row counts, V not a multiple of
    8 (no planes), T = 1, S = 1, every contraction path of the engine (batch-sized, generic, ring, split-K)."""
    from dualvgr_videoqa_amd import synth
    rng = np.random.default_rng(2024)
    for case in range(20):
        S, T = int(rng.integers(1, 90)), int(rng.integers(1, 13))
        V, H = 4 * int(rng.integers(1, 40)), 32 * int(rng.integers(1, 4))
        with_lens = bool(rng.integers(0, 2))
        want_final, want_seq = [(True, True), (True, False), (False, True)][int(rng.integers(0, 3))]
        sd_np = synth.xavier_state_dict(CT.lstm_shapes("", V, H), seed=100 + case, bias_range=0.1)
        x_np = synth.normal("sw_x", (S, T, V), 0.7, 100 + case)
        lens_np = rng.integers(1, T + 1, S).astype(np.int64)
        m = CL.BiLSTM(V, H, dtype="fp32")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
        m = m.cuda().eval()
        x = torch.from_numpy(x_np.copy()).cuda().requires_grad_(True)
        lens = torch.from_numpy(lens_np).cuda() if with_lens else None
        final, seq = m.run(x, lens, 0, want_final=want_final, want_seq=want_seq)
        Rf, Rs = synth.normal("sw_Rf", (S, 2 * H), 1.0, case), synth.normal("sw_Rs", (S, T, 2 * H), 1.0, case)
        loss = 0.0
        if want_final:
            loss = loss + (final * torch.from_numpy(Rf).cuda()).sum()
        if want_seq:
            loss = loss + (seq * torch.from_numpy(Rs).cuda()).sum()
        loss.backward()
        sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sd_np.items()}
        xo = torch.from_numpy(x_np.copy()).requires_grad_(True)
        of, os_ = CT.packed_bilstm(sd, "", xo, torch.from_numpy(lens_np) if with_lens else None)
        lo = 0.0
        if want_final:
            lo = lo + (of * torch.from_numpy(Rf)).sum()
        if want_seq:
            lo = lo + (os_ * torch.from_numpy(Rs)).sum()
        lo.backward()
        tag = (case, S, T, V, H, with_lens, want_final, want_seq)
        if want_final:
            assert rel_err(final.detach().cpu().numpy(), of.detach().numpy()) < TOL, tag
        if want_seq:
            assert rel_err(seq.detach().cpu().numpy(), os_.detach().numpy()) < TOL, tag
        assert rel_err(x.grad.cpu().numpy(), xo.grad.numpy()) < GTOL, tag
        for k, p in m.named_parameters():
            ref = sd[k].grad.numpy()
            if np.abs(ref).max() == 0.0:
                assert float(p.grad.abs().max()) == 0.0, (tag, k)
            else:
                assert rel_err(p.grad.cpu().numpy(), ref) < GTOL, (tag, k)
