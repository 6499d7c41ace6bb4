#!/usr/bin/env python3
"""End-to-end check of the whole pipeline under a real optimizer: a small synthetic VideoQA set (answer = a fixed random
function of the video and the first question token) written in the on-disk layout, read by VideoQAFlatLoader, trained with
the package's training step (trainer.TrainStep = train.py:139-159: cross-entropy + alpha * common + beta * dependence,
clip 12, Adam) for a few epochs.  The training loss has to fall and the training accuracy to rise well above chance -- every gradient path
(encoders, unit, head, auxiliary losses) takes part.  usage: python tools/train_synthetic.py [epochs]"""
import json
import os
import pickle
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dualvgr_videoqa_amd as dv

# the reference's switches (train.py:372-375) and yml keys (config.py:7-56) through dualvgr_videoqa_amd.config: a yml written for the
# reference sets clip count, widths, learning rate, batch size and seed; without --cfg the small defaults below
import argparse
_ap = argparse.ArgumentParser()
_ap.add_argument("epochs", nargs="?", type=int, default=6)
rcfg, args = dv.config.parse(parser=_ap)
epochs = args.epochs
V, N, FR, FEAT, Q, T, VQ, VA, B = 48, 8, 16, 2048, 768, 10, 200, 12, 32
LR, UNIT_LAYERS = 1e-3, 2
if args.cfg_file:
    N, FEAT, B, LR = rcfg.train["num_of_nodes"], rcfg.train["vision_dim"], min(rcfg.train["batch_size"], 64), rcfg.train["lr"]
    UNIT_LAYERS = args.unit_layers
rng = np.random.default_rng(3)
root = tempfile.mkdtemp(prefix="dvgr_train_")
ids = np.arange(1000, 1000 + V)
dv.write_flat(os.path.join(root, "a.dvgrf"), ids, np.abs(rng.standard_normal((V, N, FR, FEAT), dtype=np.float32)) * 0.5)
dv.write_flat(os.path.join(root, "m.dvgrf"), ids, np.abs(rng.standard_normal((V, N, FEAT), dtype=np.float32)) * 0.5)
vid = rng.integers(0, V, Q)
qlen = rng.integers(3, T + 1, Q)
q = rng.integers(2, VQ, (Q, T)) * (np.arange(T)[None] < qlen[:, None])
table = rng.integers(0, VA, (V, VQ))
ans = table[vid, q[:, 0]]
pickle.dump({"questions": q.astype(np.int32), "questions_len": qlen.astype(np.int32), "question_id": list(range(Q)),
             "video_ids": ids[vid], "answers": list(ans), "glove": np.zeros((VQ, 300), np.float32)}, open(os.path.join(root, "q.pt"), "wb"))
vocab = {"question_token_to_idx": {str(i): i for i in range(VQ)}, "answer_token_to_idx": {str(i): i for i in range(VA)},
         "question_answer_token_to_idx": {}}
json.dump(vocab, open(os.path.join(root, "v.json"), "w"))

torch.manual_seed(rcfg.seed if args.cfg_file else 0)
loader = dv.VideoQAFlatLoader(vocab_json=os.path.join(root, "v.json"), question_pt=os.path.join(root, "q.pt"),
                              appearance_feat=os.path.join(root, "a.dvgrf"), motion_feat=os.path.join(root, "m.dvgrf"),
                              batch_size=B, shuffle=True, seed=0)
mk = dv.config.model_kwargs(rcfg, unit_layers=UNIT_LAYERS, vocab=vocab) if args.cfg_file else dict(
    vision_dim=FEAT, module_dim=768, word_dim=300, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1, unit_layers=UNIT_LAYERS)
model = dv.DualVGR(dtype="bf16x3", **mk).cuda().train()
model.visual_input_unit.reuse_buffers = True
sk = dv.config.step_kwargs(rcfg, args.alpha, args.beta)
sk["lr"] = LR
step = dv.TrainStep(model, **sk)          # train.py:139-159 on the package's kernels
first = last = None
for ep in range(epochs):
    t0, tot, correct, n = time.perf_counter(), 0.0, 0, 0
    for _, _, answers, app, mot, question, question_len in loader:
        out = step(app, mot, question, question_len, answers)
        tot += float(out["ce"]) * len(answers)
        correct += int((out["logits"].argmax(1) == answers).sum())
        n += len(answers)
    torch.cuda.synchronize()
    print(f"epoch {ep}: cross-entropy {tot / n:.4f}  accuracy {correct / n:.3f}  ({n / (time.perf_counter() - t0):.0f} QA/s incl. optimizer and .item() syncs)")
    first = tot / n if first is None else first
    last, acc = tot / n, correct / n
assert last < 0.85 * first and acc > 2.0 / VA, (first, last, acc)
print("training check ok")
