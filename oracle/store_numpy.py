"""ORACLE (test infrastructure, not product code): numpy restatement of the reference's data path for one batch --
`VideoQADataset.__getitem__` (DataLoader.py:61-84) followed by torch's default collate as `VideoQADataLoader` uses it
(DataLoader.py:90-168) -- over in-memory arrays standing for the two `.h5` files (`resnet_features` / `resnext_features` /
`ids`, written by preprocess/preprocess_features.py:143-203).

PARITY UNPINNED: the reference's DataLoader.py imports h5py, which is absent from this image, so the reference loader
cannot be run here and no golden vectors exist for this row; the restatement below is checked by inspection against the
cited lines only.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

QUESTION_CATEGORY = {'count': 0, 'exist': 1, 'query_color': 2, 'query_size': 3, 'query_actiontype': 4, 'query_actiondir': 5,
                     'query_shape': 6, 'greater_than': 7, 'equal_to': 8, 'less_than': 9, 'equal_color': 10, 'equal_size': 11,
                     'equal_actiontype': 12, 'equal_actiondir': 13, 'equal_shape': 14}      # DataLoader.py:29-30


def id_to_index(ids):
    """DataLoader.py:139, :143: {str(id): i for i, id in enumerate(ids)} -- a later duplicate overwrites an earlier one."""
    return {str(int(v)): i for i, v in enumerate(ids)}


def dataset_item(index, obj, app_feats, app_ids, mot_feats, mot_ids):
    """One sample as `__getitem__` returns it; obj = the questions pickle dict."""
    answer = obj["answers"][index] if obj.get("answers") is not None else None               # :62
    question = np.asarray(obj["questions"])[index].astype(np.int64)                           # :64 (LongTensor)
    cat = obj.get("question_category")
    category = None
    if cat is not None:                                                                       # :65
        c = cat[index]
        category = QUESTION_CATEGORY[c] if isinstance(c, str) else int(c)
    question_len = int(np.asarray(obj["questions_len"])[index])                               # :66
    video_idx = int(np.asarray(obj["video_ids"])[index])                                      # :67
    question_idx = obj["question_id"][index]                                                  # :68
    app_index = id_to_index(app_ids)[str(video_idx)]                                          # :69
    mot_index = id_to_index(mot_ids)[str(video_idx)]                                          # :70
    appearance = app_feats[app_index]                                                         # :71-72
    motion = mot_feats[mot_index]                                                             # :73-74
    if category is None:                                                                      # :77-84
        return (video_idx, question_idx, answer, appearance, motion, question, question_len)
    return (video_idx, question_idx, category, answer, appearance, motion, question, question_len)


def collate(items):
    """torch.utils.data default_collate for these tuples: numbers -> int64 vectors, arrays -> stacked."""
    out = []
    for col in zip(*items):
        if col[0] is None:
            out.append(None)
        elif isinstance(col[0], np.ndarray):
            out.append(np.stack(col))
        else:
            out.append(np.asarray(col, dtype=np.int64))
    return tuple(out)


def batches(obj, app_feats, app_ids, mot_feats, mot_ids, batch_size, order=None, limit=0):
    n = len(obj["questions"]) if not limit else min(limit, len(obj["questions"]))
    order = np.arange(n) if order is None else np.asarray(order)
    for i in range(0, len(order), batch_size):
        yield collate([dataset_item(int(j), obj, app_feats, app_ids, mot_feats, mot_ids) for j in order[i:i + batch_size]])
