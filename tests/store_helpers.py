"""A small synthetic dataset in the reference's on-disk layout, minus HDF5: questions pickle (preprocess/datautils/svqa.py
:120-140 keys), vocab json, and the two feature tables written as flat files (dualvgr_videoqa_amd.store.write_flat)."""
import json
import os
import pickle

import numpy as np

from dualvgr_videoqa_amd import store as ST
from dualvgr_videoqa_amd import synth


def make_dataset(root, V=11, N=3, FR=4, FEAT=32, Q=23, T=6, with_category=True, seed=7):
    os.makedirs(root, exist_ok=True)
    ids = (100 + 7 * np.arange(V)).astype(np.int64)[::-1].copy()          # not sorted, not dense
    mot_ids = np.roll(ids, 3)                                             # the two files order their videos differently
    app = synth.normal("st_app", (V, N, FR, FEAT), 1.0, seed)
    mot = synth.normal("st_mot", (V, N, FEAT), 1.0, seed)
    pick = (synth.uniform("st_vid", (Q,), 0.0, 1.0, seed) * V).astype(np.int64).clip(0, V - 1)
    qlen = (1 + (synth.uniform("st_ql", (Q,), 0.0, 1.0, seed) * T).astype(np.int64)).clip(1, T)
    q = (2 + (synth.uniform("st_q", (Q, T), 0.0, 1.0, seed) * 30).astype(np.int64)) * (np.arange(T)[None] < qlen[:, None])
    names = list(ST.QUESTION_CATEGORY)
    obj = {"questions": q.astype(np.int32), "questions_len": qlen.astype(np.int32), "question_id": list(range(1000, 1000 + Q)),
           "video_ids": ids[pick], "video_names": [f"v{int(i)}" for i in ids[pick]],
           "answers": list((synth.uniform("st_ans", (Q,), 0.0, 1.0, seed) * 9).astype(np.int64)),
           "glove": synth.normal("st_glove", (32, 8), 1.0, seed)}
    if with_category:
        obj["question_category"] = [names[i % len(names)] for i in range(Q)]
    paths = {k: os.path.join(root, k) for k in ("questions.pt", "vocab.json", "app.dvgrf", "mot.dvgrf")}
    with open(paths["questions.pt"], "wb") as f:
        pickle.dump(obj, f)
    vocab = {"question_token_to_idx": {f"w{i}": i for i in range(32)}, "answer_token_to_idx": {f"a{i}": i for i in range(9)},
             "question_answer_token_to_idx": {f"w{i}": i for i in range(32)}}
    with open(paths["vocab.json"], "w") as f:
        json.dump(vocab, f)
    ST.write_flat(paths["app.dvgrf"], ids, app)
    ST.write_flat(paths["mot.dvgrf"], mot_ids, mot)
    mot_by_file = mot                                                      # row i of the motion file belongs to mot_ids[i]
    return obj, (app, ids), (mot_by_file, mot_ids), paths
