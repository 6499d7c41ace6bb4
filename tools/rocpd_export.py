#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this turns one into the two text forms the rest of the
tooling reads:

    tools/rocpd_export.py stats   <results.db> <out.csv> [<out_summary.txt>]   # --kernel-trace --stats run
    tools/rocpd_export.py counters <results.db> <out.csv>                      # --pmc run -> Kernel_Name,Counter_Name,Counter_Value

`stats` lists every kernel with calls, total / average / min / max duration (ns -> us) and share of the summed kernel time,
the same columns as rocprofv3's kernel_stats.csv."""
import csv
import sqlite3
import sys


def stats(db, out_csv, out_txt=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    # the one-off stream-placement probe (csrc/dvgr_place.hip: k_place_hold / k_place_link / k_place_tiny, ~30 ms at set-up) is
    # listed apart: it is not part of a step
    probe = [r for r in rows if "k_place_" in r[0]]
    rows = [r for r in rows if "k_place_" not in r[0]]
    total = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, k, s, a, lo, hi in rows:
            w.writerow([n, k, s, f"{a:.1f}", f"{100.0 * s / total:.2f}", lo, hi])
    if out_txt:
        with open(out_txt, "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats, {db}\n")
            f.write(f"{'kernel':100s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share%':>7s}\n")
            for n, k, s, a, lo, hi in rows:
                f.write(f"{n[:100]:100s} {k:7d} {a / 1e3:9.2f} {lo / 1e3:9.2f} {hi / 1e3:9.2f} {100.0 * s / total:7.2f}\n")
            f.write(f"summed kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} launches\n")
            if probe:
                f.write(f"(not counted: stream-placement probe at set-up, {sum(r[2] for r in probe) / 1e6:.3f} ms in "
                        f"{sum(r[1] for r in probe)} launches of k_place_*)\n")


def counters(db, out_csv):
    c = sqlite3.connect(db)
    cur = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id")
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for r in cur:
            w.writerow(r)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:5])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
