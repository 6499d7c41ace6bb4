#!/usr/bin/env python3
"""Fold the A/B rows a GPU call brought back (gpurun_out/ab_rows/*.tsv, written by tools/ab.py) into
profiles/experiments_r06.tsv; rows already present (same experiment, variant, build, numbers) are skipped."""
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles", "experiments_r06.tsv")
hdr = "date\texperiment\tvariant\tlib\tenv\tbuild\tmedian_ms\tmin_ms\trounds\tlaunches_per_step\tbench_args\n"
lines = open(out).read().splitlines(True) if os.path.exists(out) else [hdr]
seen = {tuple(l.split("\t")[1:3] + l.split("\t")[5:8]) for l in lines[1:]}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "ab_rows", "*.tsv"))):
    for l in open(f):
        if not l[:13] >= "2026-10-04T21":          # (gpurun_out/ carries rows of earlier rounds as well: round 5 began 21:00Z)
            continue
        k = tuple(l.split("\t")[1:3] + l.split("\t")[5:8])
        if k not in seen:
            seen.add(k)
            lines.append(l)
open(out, "w").write("".join(lines))
print(len(lines) - 1, "rows")
