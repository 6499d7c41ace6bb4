"""GPU: the whole-model wrapper `DualVGR` (encoders, unit and answer head on HIP; one parametrisation swaps the stock
nn.LSTM encoders of tools/stock_encoders.py in, to keep that A/B timing aid honest) against the logits
and auxiliary outputs the imported reference model produced for the same seeded weights and inputs
(tests/golden/g_model_small.npz, make_golden_model.py).  A reference state_dict loads with strict=True."""
import os

import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from dualvgr_videoqa_amd import synth
from helpers import rel_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_model_small.npz")


def seeded_state(keys, shapes, seed):
    """As tests/golden/make_golden_model.py::seeded_state."""
    sd = {}
    for k, sh in zip(keys, shapes):
        shape = tuple(int(x) for x in sh.strip("()").split(",") if x.strip())
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(1.0 + 0.3 * synth.uniform(k, shape, 0.0, 1.0, seed))
        elif len(shape) >= 2:
            bound = (6.0 / (shape[0] + shape[-1])) ** 0.5
            sd[k] = torch.from_numpy(synth.uniform(k, shape, -bound, bound, seed))
        else:
            sd[k] = torch.from_numpy(synth.uniform(k, shape, -0.1, 0.1, seed))
            if k.endswith("classifier.3.weight"):
                sd[k] = sd[k] + 1.0
    return sd


def model_inputs(B, N, Fr, V, T, vocab_q, seed):
    app = np.abs(synth.normal("m_app", (B, N, Fr, V), 0.5, seed))
    mot = np.abs(synth.normal("m_mot", (B, N, V), 0.5, seed))
    qlen = (3 + (synth.uniform("m_qlen", (B,), 0.0, 1.0, seed) * (T - 2)).astype(np.int64)).clip(1, T)
    qlen[0] = T
    q = (2 + (synth.uniform("m_q", (B, T), 0.0, 1.0, seed) * (vocab_q - 2)).astype(np.int64)).clip(2, vocab_q - 1)
    q = q * (np.arange(T)[None, :] < qlen[:, None])
    return app.astype(np.float32), mot.astype(np.float32), q.astype(np.int64), qlen.astype(np.int64)


@pytest.mark.parametrize("appearance_encoder,question_encoder,dtype", [("hip", "hip", "fp32"), ("hip", "hip", "bf16x3"),
                                                                       ("miopen", "miopen", "fp32")])
def test_whole_model_matches_reference_golden_and_trains(appearance_encoder, question_encoder, dtype):
    z = np.load(GOLDEN)
    V, D, Wd, N, Fr, L, B, T, vq, va, seed = [int(v) for v in z["meta"]]
    vocab = {"question_token_to_idx": {str(i): i for i in range(vq)}, "answer_token_to_idx": {str(i): i for i in range(va)}}
    m = dv.DualVGR(vision_dim=V, module_dim=D, word_dim=Wd, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1,
                   unit_layers=L, dtype=dtype)
    if "miopen" in (appearance_encoder, question_encoder):
        from tools.stock_encoders import use_stock_encoders
        use_stock_encoders(m, appearance=appearance_encoder == "miopen", question=question_encoder == "miopen")
    keys = [str(k) for k in z["keys"]]
    assert sorted(m.state_dict().keys()) == keys                      # the reference's state_dict surface, key for key
    m.load_state_dict(seeded_state(keys, [str(s) for s in z["shapes"]], seed), strict=True)
    m = m.cuda().eval()
    app, mot, q, qlen = model_inputs(B, N, Fr, V, T, vq, seed)
    outs = m(torch.from_numpy(app).cuda(), torch.from_numpy(mot).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(qlen).cuda())
    logits, aq_embed, mq_embed, com_app, com_mot, aq_fusion, mq_fusion = outs
    assert rel_err(logits.detach().cpu().numpy(), z["logits"]) < 1e-4
    assert rel_err(aq_embed.detach().cpu().numpy(), z["aq_embed"]) < 1e-4
    assert rel_err(mq_embed.detach().cpu().numpy(), z["mq_embed"]) < 1e-4
    assert rel_err(com_app[-1].detach().cpu().numpy(), z["com_app_last"]) < 1e-4
    assert rel_err(mq_fusion[-1].detach().cpu().numpy(), z["mq_fusion_last"]) < 1e-4
    assert (logits.argmax(1).cpu().numpy() == z["logits"].argmax(1)).all()
    # one optimizer step in training mode end to end (encoders, unit, head): finite loss, every parameter gets a gradient
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    target = torch.randint(0, va, (B,), device="cuda")
    out = m(torch.from_numpy(app).cuda(), torch.from_numpy(mot).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(qlen).cuda())
    loss = torch.nn.functional.cross_entropy(out[0], target) + 1e-3 * sum(x.pow(2).mean() for x in out[3] + out[5])
    loss.backward()
    assert torch.isfinite(loss)
    missing = [k for k, p in m.named_parameters() if p.grad is None]
    assert not missing, missing
    opt.step()
