"""GPU: the batch loader (row f4) -- `VideoQAFlatLoader` against the numpy restatement of the reference's
`VideoQADataset.__getitem__` + default collate (oracle/store_numpy.py), batch by batch, byte-exact (it is a copy path)."""
import numpy as np
import pytest
import torch

import dualvgr_videoqa_amd as dv
from oracle import store_numpy as SN
from store_helpers import make_dataset

pytestmark = pytest.mark.gpu


def loader_kwargs(paths, **kw):
    d = dict(vocab_json=paths["vocab.json"], question_pt=paths["questions.pt"], appearance_feat=paths["app.dvgrf"],
             motion_feat=paths["mot.dvgrf"], batch_size=4)
    d.update(kw)
    return d


def check_batch(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w is None:
            assert g is None
            continue
        g = g.cpu().numpy() if torch.is_tensor(g) else np.asarray(g)
        assert g.shape == w.shape and np.array_equal(g, w)


@pytest.mark.parametrize("with_category", [True, False])
def test_loader_batches_equal_the_reference_collation(tmp_path, with_category):
    obj, (app, ids), (mot, mot_ids), paths = make_dataset(str(tmp_path), with_category=with_category)
    ld = dv.VideoQAFlatLoader(**loader_kwargs(paths, num_workers=4, pin_memory=True))
    want = list(SN.batches(obj, app, ids, mot, mot_ids, batch_size=4))
    assert len(ld) == len(want) == 6                                       # ceil(23 / 4), DataLoader.py:167-168
    assert ld.vocab["answer_idx_to_token"][3] == "a3" and ld.glove_matrix.shape == (32, 8) and ld.batch_size == 4
    n = 0
    for got, w in zip(ld, want):
        assert got[-4].is_cuda and got[-3].is_cuda and got[-2].dtype == torch.int64
        check_batch(got, w)
        n += 1
    assert n == 6
    for got, w in zip(ld, want):                                           # a second epoch yields the same batches
        check_batch(got, w)


def test_subset_shuffle_and_drop_last(tmp_path):
    obj, (app, ids), (mot, mot_ids), paths = make_dataset(str(tmp_path))
    ld = dv.VideoQAFlatLoader(**loader_kwargs(paths, train_num=10, batch_size=3, shuffle=True, seed=5, drop_last=True))
    assert len(ld) == 3
    order = np.concatenate(ld.batch_indices())
    assert sorted(order.tolist()) != order.tolist() and set(order.tolist()) <= set(range(10)) and len(order) == 9
    ld._epoch = 0                                                          # replay the first epoch's permutation
    want = list(SN.batches(obj, app, ids, mot, mot_ids, batch_size=3, order=order))
    for got, w in zip(ld, want):
        check_batch(got, w)


def test_early_exit_and_model_sized_rows(tmp_path):
    """Break out of an epoch (the producer thread must stop), and rows of the real size (N=20, 16 x 2048: 2.6 MB each)."""
    rng = np.random.default_rng(1)
    V, Q = 6, 12
    app = rng.standard_normal((V, 20, 16, 2048)).astype(np.float32)
    mot = rng.standard_normal((V, 20, 2048)).astype(np.float32)
    ids = np.arange(10, 10 + V)
    import json, os, pickle
    root = str(tmp_path)
    dv.write_flat(os.path.join(root, "a.dvgrf"), ids, app)
    dv.write_flat(os.path.join(root, "m.dvgrf"), ids, mot)
    obj = {"questions": rng.integers(1, 50, (Q, 7)).astype(np.int32), "questions_len": np.full(Q, 7, np.int32),
           "question_id": list(range(Q)), "video_ids": ids[rng.integers(0, V, Q)], "answers": list(range(Q)), "glove": np.zeros((50, 4))}
    pickle.dump(obj, open(os.path.join(root, "q.pt"), "wb"))
    json.dump({"question_token_to_idx": {}, "answer_token_to_idx": {}, "question_answer_token_to_idx": {}}, open(os.path.join(root, "v.json"), "w"))
    ld = dv.VideoQAFlatLoader(vocab_json=os.path.join(root, "v.json"), question_pt=os.path.join(root, "q.pt"),
                              appearance_feat=os.path.join(root, "a.dvgrf"), motion_feat=os.path.join(root, "m.dvgrf"), batch_size=5)
    want = list(SN.batches(obj, app, ids, mot, ids, batch_size=5))
    for got, w in zip(ld, want):
        check_batch(got, w)
        break
    for got, w in zip(ld, want):
        check_batch(got, w)
