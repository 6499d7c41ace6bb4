#!/usr/bin/env python3
"""Converts the reference's feature files (preprocess/preprocess_features.py:143-203: `<name>_appearance_feat.h5` with
datasets `resnet_features (V, N, 16, 2048)` + `ids (V,)`, `<name>_motion_feat.h5` with `resnext_features (V, N, 2048)` +
`ids`) into the flat format of include/dvgr_store.h, in chunks (the appearance file of SVQA is tens of GB).

    python tools/h5_to_flat.py svqa_appearance_feat.h5 svqa_appearance_feat.dvgrf
    python tools/h5_to_flat.py svqa_motion_feat.h5 svqa_motion_feat.dvgrf

Needs h5py (not part of the build image: run it where the .h5 files were produced)."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualvgr_videoqa_amd.store import MAGIC  # noqa: E402


def convert(src, dst, chunk_items=64):
    try:
        import h5py
    except ImportError:
        raise SystemExit("h5py is required to read " + src)
    with h5py.File(src, "r") as f:
        name = "resnet_features" if "resnet_features" in f else "resnext_features"
        feats, ids = f[name], np.asarray(f["ids"][()], dtype=np.int64)
        item = tuple(feats.shape[1:])
        shape = list(item) + [1] * (4 - len(item))
        ids_off = 80
        data_off = (ids_off + 8 * len(ids) + 4095) // 4096 * 4096
        with open(dst, "wb") as out:
            out.write(struct.pack("<8sIIII4QQQQ", MAGIC, 1, 0, len(item), 0, *shape, len(ids), ids_off, data_off))
            out.write(ids.tobytes())
            out.write(b"\0" * (data_off - ids_off - 8 * len(ids)))
            for i in range(0, len(ids), chunk_items):
                out.write(np.ascontiguousarray(feats[i:i + chunk_items], dtype=np.float32).tobytes())
    print(f"{dst}: {len(ids)} items of shape {item}")


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    convert(sys.argv[1], sys.argv[2])
