#!/usr/bin/env python3
"""One training step of bench.py as a two-queue timeline, from a rocprofv3 --kernel-trace CSV.
usage: trace_step.py <kernel_trace.csv> [steps-from-the-end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marker = 'k_sum_and_ones' if any('k_sum_and_ones' in r['Kernel_Name'] for r in rows) else 'k_split_planes'
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]['Start_Timestamp'])
qs = sorted({r['Queue_Id'] for r in rows[a:b]})
busy = {q: 0 for q in qs}
for r in rows[a:b]:
    n = r['Kernel_Name'].replace('void dvgr::', '').replace('dvgr::', '')
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    busy[r['Queue_Id']] += e - s
    col = qs.index(r['Queue_Id'])
    print(f"{s:8.1f} {e:8.1f} {e - s:6.1f} {'    ' * col}q{r['Queue_Id']} {n[:56]}")
print("span us:", (int(rows[b]['Start_Timestamp']) - t0) / 1e3, "busy per queue:", {q: round(v, 1) for q, v in busy.items()})
