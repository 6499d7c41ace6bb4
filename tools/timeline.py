#!/usr/bin/env python3
"""One training step's kernel timeline from a rocprofv3 --kernel-trace database (rocpd): start / end / duration / queue.

    tools/timeline.py <results.db> [step_index] [anchor-kernel-substring]

A step runs from one launch of the anchor kernel to the next; the default anchor (k_split_planes) is the unit's first launch,
the whole model (bench.py --model) launches that kernel several times per step: anchor on k_words_fwd there."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_split_planes"
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
for r in rows[i0:i1]:
    n = r[0].replace("dvgr::", "").replace("void ", "")[:60]
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} q{r[3]} {n}")
print(f"step period {(rows[i1][1] - t0) / 1e3:.1f} us")
