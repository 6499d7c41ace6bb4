#!/usr/bin/env python3
"""Summarises an AMD_LOG_LEVEL=3/4 log of the HIP runtime: per API name the number of calls and the host time between the
entry line and its 'Returned' line (same thread), and the most frequent message templates of level 4 (signal waits, barrier
packets, queue switches).  Used to see WHERE the host spends the extra enqueue time in the slow stream set-ups of
profiles/HISTORY.md section 5.   usage: python tools/amdlog_hist.py LOG [skip_fraction=0.5]"""
import collections
import re
import sys

ansi = re.compile(r"\x1b\[[0-9;]*m")
line_re = re.compile(r"^:(\d):(\S+)\s*:(\d+)\s*: (\d+) us:\s*(?:\[pid:(\d+)\s+tid:\s*(0x[0-9a-f]+)\])?\s*(.*)$")
num = re.compile(r"0x[0-9a-fA-F]+|\b\d+\b")


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    lines = open(path, errors="replace").read().splitlines()
    lines = lines[int(len(lines) * skip):]                  # the steady state, not the set-up
    open_call = {}
    dur = collections.defaultdict(lambda: [0, 0.0, 0.0])
    templ = collections.Counter()
    t_first = t_last = None
    for raw in lines:
        m = line_re.match(ansi.sub("", raw))
        if not m:
            continue
        lvl, src, _, ts, _, tid, msg = m.groups()
        ts = int(ts)
        t_first = ts if t_first is None else t_first
        t_last = ts
        msg = msg.strip()
        ent = re.match(r"^(hip\w+) \(", msg)
        ret = re.match(r"^(hip\w+): Returned (\w+)", msg)
        if ent:
            open_call[tid] = (ent.group(1), ts)
        elif ret and tid in open_call and open_call[tid][0] == ret.group(1):
            name, t0 = open_call.pop(tid)
            d = dur[name]
            d[0] += 1
            d[1] += ts - t0
            d[2] = max(d[2], ts - t0)
        else:
            templ[(lvl, src, num.sub("#", msg)[:110])] += 1
    print(f"{len(lines)} lines, {0 if t_first is None else (t_last - t_first) / 1e3:.1f} ms of log")
    print("API                                   calls   total ms   mean us    max us")
    for name, (n, tot, mx) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{name:36s} {n:6d} {tot / 1e3:10.2f} {tot / max(n, 1):9.1f} {mx:9.0f}")
    print("most frequent other messages:")
    for (lvl, src, t), n in templ.most_common(40):
        print(f"{n:7d}  :{lvl}:{src}: {t}")


if __name__ == "__main__":
    main()
