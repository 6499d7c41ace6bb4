"""GPU: the training step around the path: `dualvgr_videoqa_amd.TrainStep` against
  (1) two iterations of the IMPORTED reference's loop, train.py:139-159, recorded in tests/golden/g_step_small.npz
      (make_golden_step.py: training-mode BatchNorm, dropout probabilities patched to 0): loss and its terms, the total
      gradient norm of clip_grad_norm_, every parameter and BatchNorm buffer after the last step;
  (2) torch.nn.utils.clip_grad_norm_ + torch.optim.Adam run on the device for the fused clip + Adam launches alone."""
import os

import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import synth, trainer
from test_hip_model import model_inputs, seeded_state

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_step_small.npz")


@pytest.mark.parametrize("case", ["a/", "b/"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_two_training_steps_match_the_reference_loop(case, dtype):
    z = np.load(GOLDEN)
    V, D, Wd, N, Fr, L, B, T, vq, va, seed, steps = [int(v) for v in z[case + "meta"]]
    lr, alpha, beta, max_norm, a_scale = [float(v) for v in z[case + "meta_f"]]
    vocab = {"question_token_to_idx": {str(i): i for i in range(vq)}, "answer_token_to_idx": {str(i): i for i in range(va)}}
    m = dv.DualVGR(vision_dim=V, module_dim=D, word_dim=Wd, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1,
                   unit_layers=L, dtype=dtype)
    keys = [str(k) for k in z[case + "keys"]]
    before = seeded_state(keys, [str(s) for s in z[case + "shapes"]], seed)
    for k in before:                             # make_golden_step.step_state: logits that straddle the LeakyReLU kink
        if k.endswith(".a.weight"):
            before[k] = before[k] * a_scale
    m.load_state_dict(before, strict=True)
    m = m.cuda().train()
    m.set_dropout_enabled(False)                 # the fixture was recorded with p = 0 at every site
    app, mot, q, qlen = [torch.from_numpy(x).cuda() for x in model_inputs(B, N, Fr, V, T, vq, seed)]
    answers = torch.from_numpy((synth.uniform("s_ans", (B,), 0.0, float(va), seed)).astype(np.int64)).cuda()
    step = dv.TrainStep(m, lr=lr, alpha=alpha, beta=beta, max_norm=max_norm)
    noise = {}                                   # per parameter: elements whose reference gradient is rounding noise
    p_dim = {k: p.dim() for k, p in m.named_parameters()}
    for i in range(steps):
        out = step(app, mot, q, qlen, answers)
        got = {k: float(out[k]) for k in ("loss", "ce", "common", "dependence")}
        norm = float(out["grad_sqnorm"]) ** 0.5
        # the loss, its three terms and clip_grad_norm_'s total norm against the reference loop's, EVERY step.  Step 0 sees
        # the reference's weights exactly; step 1 sees weights after one Adam step (noise elements moved by +-lr in either
        # implementation, see below), so it gets a wider rtol.
        want = {"loss": z[case + "loss"][i], "ce": z[case + "ce"][i], "common": z[case + "common"][i], "dependence": z[case + "dep"][i]}
        rt = (2e-4 if dtype == "fp32" else 5e-4) if i == 0 else 5e-3
        rels = {k: abs(got[k] - float(want[k])) / max(abs(float(want[k])), 1e-12) for k in want}
        rels["norm"] = abs(norm - float(z[case + "norm"][i])) / float(z[case + "norm"][i])
        print(f"  {case} {dtype} step {i}: loss {got['loss']:.6f} (ref {float(want['loss']):.6f}), relative errors "
              + ", ".join(f"{k} {v:.1e}" for k, v in rels.items()))
        assert max(rels.values()) < rt, (i, rels)
        if i == 0:
            # the gradients themselves (the step wrote them back multiplied by the clip coefficient)
            coef = min(1.0, max_norm / (norm + 1e-6))
            gerrs = {}
            gmax_all = max(float(np.abs(z["a/g0/" + k]).max()) for k, _ in m.named_parameters())
            for k, p in m.named_parameters():
                want = z["a/g0/" + k]               # (both cases start from the same weights and batch)
                have = (p.grad.detach().float().cpu().numpy().reshape(-1)[:2048] / coef) if p.grad is not None else np.zeros_like(want)
                scale = max(float(np.abs(want).max()), 1e-6)
                noise[k] = (np.abs(want) < 0.02 * scale) | (np.abs(want) < 1e-6 * gmax_all)
                gerrs[k] = float(np.abs(have - want).max()) / max(scale, 1e-3)
            top = sorted(gerrs.items(), key=lambda kv: -kv[1])[:4]
            print(f"  {case} {dtype} step 0: worst gradient errors {[(k, f'{v:.2e}') for k, v in top]} (max |err| / max |g| per tensor)")
            # weight matrices: the 5e-4 gradient gate.  Vectors (biases, BatchNorm, a.bias) are sums over rows that cancel:
            # in exact fp32 they meet the same gate, in bf16x3 (1e-5 operand rounding) they get 10 x
            mats = [v for k, v in gerrs.items() if p_dim[k] >= 2]
            vecs = [v for k, v in gerrs.items() if p_dim[k] < 2]
            assert max(mats) < 5e-4, top
            assert max(vecs) < (5e-4 if dtype == "fp32" else 5e-3), top
    torch.cuda.synchronize()
    # every parameter / buffer after the last step.  Adam's first steps move each element by ~lr * sign(g): an element whose
    # gradient is rounding noise around an analytic zero (a constant added to every logit of a softmax; the bias of a
    # feature that all 6 samples put on the linear side of the ELU in front of BatchNorm; ...) moves by +-lr in either
    # implementation, and an element whose gradient is small against its tensor's largest is known only to 5e-4 of that
    # largest (the gradient gate above), which Adam's normalisation turns into a large relative change of its update.  So
    # elements whose reference gradient is below 2 % of its tensor's largest are left out and the rest is measured in
    # units of lr.  (The optimiser's own arithmetic is held to torch's, to 2e-6, in the next test.)
    sd = m.state_dict()
    worst = {}
    for k in keys:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == steps
            continue
        if case + "after/" + k not in z.files:
            continue
        got = sd[k].detach().float().cpu().numpy().reshape(-1)
        want = z[case + "after/" + k]
        moved = np.abs(want - before[k].numpy().reshape(-1)[:2048])
        err = np.abs(got[:2048] - want)
        if k in noise:
            keep = ~noise[k]
            if not keep.any():
                continue
            err, moved = err[keep], moved[keep]
        tol = 0.1 * lr * steps if not k.endswith(("running_mean", "running_var")) else 2e-4 * max(1.0, float(np.abs(want).max()))
        worst[k] = (float(err.max()), float(moved.max()))
        frac_bad = float((err > tol).mean())
        assert frac_bad <= 2e-2, (k, frac_bad, float(err.max()), tol)
        # hard cap: no kept element may be off by more than Adam can move it in `steps` steps (|update| <= ~lr per step)
        if not k.endswith(("running_mean", "running_var")):
            assert float(err.max()) <= 2.2 * lr * steps, (k, float(err.max()))
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:4]
    print("  worst |err| (|moved|):", [(k, f"{e:.2e} ({mv:.2e})") for k, (e, mv) in top])
    assert any(mv > 0.5 * lr for _, (e, mv) in worst.items())          # the step really moved the weights


@pytest.mark.parametrize("max_norm", [12.0, 0.5, 0.0])
def test_fused_clip_and_adam_match_torch(max_norm):
    """Three segments of awkward sizes (vector path, scalar tail, a 1-element bias), four steps, against
    clip_grad_norm_ + torch.optim.Adam (foreach=False) on the same device."""
    g = torch.Generator().manual_seed(3)
    sizes = [4096 * 37 + 3, 1, 1028, 5_000_000]
    ps = [torch.randn(n, generator=g).cuda() for n in sizes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=3e-3, foreach=False)
    mine = trainer.ClipAdam(lr=3e-3, max_norm=max_norm)
    for it in range(4):
        grads = [(torch.randn(n, generator=g) * (10.0 if it % 2 else 0.01)).cuda() for n in sizes]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        want_norm = float(torch.nn.utils.clip_grad_norm_(ref, max_norm, foreach=False)) if max_norm > 0 else \
            float(torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(gr) for gr in grads])))
        opt.step()
        sq = mine.step([(i, p, gr) for i, (p, gr) in enumerate(zip(ps, grads))])
        assert abs(float(sq) ** 0.5 - want_norm) < 1e-5 * want_norm
        for p, r, gr in zip(ps, ref, grads):
            assert float((p - r.detach()).abs().max()) < 2e-6 * max(1.0, float(r.abs().max()))
            if max_norm > 0:
                assert torch.allclose(gr, r.grad, rtol=1e-5, atol=1e-9)      # the clipped gradient is written back


def test_moments_restored_on_the_host_are_moved_not_reset():
    """A checkpoint loaded with map_location='cpu' hands ClipAdam host moments: the next step must continue from them
    (round 2 silently replaced them with zeros while keeping the step count)."""
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(10_000, generator=g).cuda()
    grads = [torch.randn(10_000, generator=g).cuda() for _ in range(3)]
    a = trainer.ClipAdam(lr=1e-2)
    pa = p0.clone()
    for gr in grads[:2]:
        a.step([("w", pa, gr.clone())])
    sd = a.state_dict()
    sd["state"] = {k: (m.cpu(), v.cpu()) for k, (m, v) in sd["state"].items()}      # as torch.load(..., map_location='cpu')
    b = trainer.ClipAdam()
    b.load_state_dict(sd)
    pb = pa.clone()
    pz = pa.clone()
    a.step([("w", pa, grads[2].clone())])
    b.step([("w", pb, grads[2].clone())])
    # (the gradient norm is summed with float atomics: the clip coefficient may differ in its last bit between two launches)
    assert b.step_count == 3 and float((pa - pb).abs().max()) <= 1e-6 * float(pa.abs().max())
    zero = trainer.ClipAdam(lr=1e-2)             # what round 2 did: moments silently reset, step count kept
    zero.step_count = 2
    zero.step([("w", pz, grads[2].clone())])
    assert float((pz - pa).abs().max()) > 1e-4   # ... which is a visibly different third step
    with pytest.raises(RuntimeError, match="elements"):
        b.step([("w", torch.zeros(5, device="cuda"), torch.zeros(5, device="cuda"))])


def test_adam_pass_keeps_the_units_bf16_planes_in_step():
    """The unit's contraction kernels read bf16 hi / lo planes of the weights; TrainStep's Adam pass rewrites them from the
    updated values, so the next forward equals a forward after an explicit repack."""
    z = np.load(GOLDEN)
    V, D, Wd, N, Fr, L, B, T, vq, va, seed, steps = [int(v) for v in z["a/meta"]]
    vocab = {"question_token_to_idx": {str(i): i for i in range(vq)}, "answer_token_to_idx": {str(i): i for i in range(va)}}
    m = dv.DualVGR(vision_dim=V, module_dim=D, word_dim=Wd, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1,
                   unit_layers=L, dtype="bf16x3").cuda().train()
    m.set_dropout_enabled(False)
    app, mot, q, qlen = [torch.from_numpy(x).cuda() for x in model_inputs(B, N, Fr, V, T, vq, seed)]
    answers = torch.zeros(B, dtype=torch.int64, device="cuda")
    step = dv.TrainStep(m, lr=1e-2)
    step(app, mot, q, qlen, answers)
    m.eval()
    with torch.no_grad():
        a = m(app, mot, q, qlen)[0].clone()
        m.visual_input_unit.params_changed()            # force dvgr_pack_weights from the fp32 parameters
        b = m(app, mot, q, qlen)[0]
    assert torch.equal(a, b)


def test_kept_planes_tensor_needs_planes_rewritten_and_gets_the_right_gradients_with_it():
    """unit.lowp_planes() contract: a caller that keeps the planes tensor and rewrites it on later steps must call
    planes_rewritten(), or the fused backward contracts dz against LAST step's transposed P1 planes.  With the call the gradients
    after two optimizer steps equal those of a fresh module holding the same parameters (planes and P1^T packed from scratch)."""
    from dualvgr_videoqa_amd import synth
    from oracle import unit_torch as O       # (shapes and the scalar loss only)
    B, N, T, D, Wd, Lr = 8, 20, 9, 768, 300, 2
    sd_np = synth.xavier_state_dict(O.param_shapes(Wd, D, Lr), seed=91, bias_range=0.05)
    inp = synth.unit_inputs(B, N, T, D, Wd, seed=91)

    def fwd_bwd(m):
        t = {k: torch.from_numpy(v.copy()).cuda() for k, v in inp.items()}
        for k in ("A", "M", "dq", "words"):
            t[k].requires_grad_(True)
        if m.last_flat_grad is not None:
            m.zero_grad_flat()
        O.bench_scalar_loss(m(t["A"], t["M"], t["dq"], t["words"], t["qlen"]), N).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}, t["A"].grad.detach().clone()

    m = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.cuda().eval()
    m.fused_gat_bwd = "force"
    fwd_bwd(m)                                              # packs the planes and P1^T once
    planes = m.lowp_planes()                                # kept across steps
    opt = trainer.ClipAdam(lr=5e-2, max_norm=12.0)
    for _ in range(2):
        opt.step([("unit", m.flat_params(), m.last_flat_grad, planes)])
        m.planes_rewritten()
    got_w, got_a = fwd_bwd(m)
    ref = dv.DualVGRUnit_multiple(word_dim=Wd, module_dim=D, num_of_nodes=N, unit_layers=Lr, dtype="bf16x3")
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    ref = ref.cuda().eval()
    ref.fused_gat_bwd = "force"
    want_w, want_a = fwd_bwd(ref)
    assert float((got_a - want_a).abs().max()) <= 2e-6 * float(want_a.abs().max())
    for k in want_w:
        if k.endswith(".bias"):
            continue
        assert float((got_w[k] - want_w[k]).abs().max()) <= 1e-5 * float(want_w[k].abs().max()), k
