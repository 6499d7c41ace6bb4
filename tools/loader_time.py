#!/usr/bin/env python3
"""Throughput of the batch loader (row f4) on a synthetic store of real-sized rows: GB/s of features delivered into HBM and
QA pairs/s, alone and while the whole-model step consumes the batches.  usage: python tools/loader_time.py [V B N]"""
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

a = sys.argv[1:]
V, B, N = [int(v) for v in (a[:3] if len(a) >= 3 else (1024, 64, 20))]
T, Q = 15, 64 * 64
root = tempfile.mkdtemp(prefix="dvgr_store_")
rng = np.random.default_rng(0)
ids = np.arange(V)
dv.write_flat(os.path.join(root, "a.dvgrf"), ids, np.abs(rng.standard_normal((V, N, 16, 2048), dtype=np.float32)) * 0.5)
dv.write_flat(os.path.join(root, "m.dvgrf"), ids, np.abs(rng.standard_normal((V, N, 2048), dtype=np.float32)) * 0.5)
obj = {"questions": rng.integers(2, 4000, (Q, T)).astype(np.int32), "questions_len": np.full(Q, T, np.int32),
       "question_id": list(range(Q)), "video_ids": ids[rng.integers(0, V, Q)], "answers": list(rng.integers(0, 1000, Q)),
       "glove": np.zeros((4000, 4), np.float32)}
pickle.dump(obj, open(os.path.join(root, "q.pt"), "wb"))
json.dump({"question_token_to_idx": {str(i): i for i in range(4000)}, "answer_token_to_idx": {str(i): i for i in range(1000)},
           "question_answer_token_to_idx": {}}, open(os.path.join(root, "v.json"), "w"))
ld = dv.VideoQAFlatLoader(vocab_json=os.path.join(root, "v.json"), question_pt=os.path.join(root, "q.pt"),
                          appearance_feat=os.path.join(root, "a.dvgrf"), motion_feat=os.path.join(root, "m.dvgrf"), batch_size=B,
                          shuffle=True, seed=1)
bytes_per_batch = B * (N * 16 * 2048 + N * 2048) * 4
for _ in ld:      # warm the page cache and the pinned buffers
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for batch in ld:
    n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"store: {V} videos x {N} clips ({V * N * 17 * 2048 * 4 / 2**30:.1f} GiB), batch {B}")
print(f"loader alone       : {n * B / dt:9.0f} QA/s  {n * bytes_per_batch / dt / 1e9:6.1f} GB/s into HBM  ({dt / n * 1e3:.2f} ms / batch)")
if os.environ.get("LOADER_WITH_MODEL", "1") == "1":
    vocab = ld.vocab
    m = dv.DualVGR(vision_dim=2048, module_dim=768, word_dim=300, vocab=vocab, num_of_nodes=N, graph_module="GAT", graph_layers=1,
                   unit_layers=3, dtype="bf16x3").cuda().train()

    def step(batch):
        _, _, ans, app, mot, q, ql = batch
        m.zero_grad(set_to_none=True)
        out = m(app, mot, q, ql)
        torch.nn.functional.cross_entropy(out[0], ans).backward()

    for i, batch in enumerate(ld):
        step(batch)
        if i == 5:
            break
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in ld:
        step(batch)
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"loader + model step: {n * B / dt:9.0f} QA/s  {n * bytes_per_batch / dt / 1e9:6.1f} GB/s into HBM  ({dt / n * 1e3:.2f} ms / step)")
import shutil
shutil.rmtree(root, ignore_errors=True)
