#!/usr/bin/env python3
"""One training step's kernel timeline from a rocprofv3 --kernel-trace database (rocpd): start / end / duration / queue.

    tools/timeline.py <results.db> [step_index]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_split_planes" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
for r in rows[i0:i1]:
    n = r[0].replace("dvgr::", "").replace("void ", "")[:60]
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} q{r[3]} {n}")
print(f"step period {(rows[i1][1] - t0) / 1e3:.1f} us")
