#!/usr/bin/env python3
"""Mean PMC counter values per kernel from a rocprofv3 --pmc counter_collection CSV.  usage: pmc_summary.py <csv> [name-filter]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if flt in n:
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n[:100], "x", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
