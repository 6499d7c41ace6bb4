#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace --stats CSV.  usage: prof_summary.py <kernel_stats.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
print(f"{'kernel':80s} calls/step  avg_us  ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    print(f"{r['Name'][:80]:80s} {int(r['Calls']) / steps:9.1f} {float(r['AverageNs']) / 1e3:7.1f} {ms:8.3f}")
print("all kernels ms/step:", round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps, 3))
