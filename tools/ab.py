#!/usr/bin/env python3
"""A/B (or A/B/C/...) of library builds and/or environment switches on ONE device: the variants' bench.py runs alternate
round-robin (devices of the pool differ by several per cent, so only same-device alternation is evidence), and one TSV row
per variant is appended to profiles/experiments_r06.tsv: experiment, variant, lib, env, build id of the sources, median /
min ms per step over the rounds, launches per step, bench arguments.

    tools/ab.py <experiment-name> [--rounds 3] [--bench "--config c2"] <variant> [<variant> ...]
    variant = name[:lib=<path.so>][:ENV=VALUE]...       e.g.  base:DVGR_SCHED=0  split:DVGR_SCHED=3
"""
import argparse
import datetime
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--bench", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "experiments_r06.tsv"))
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    variants = []
    for v in a.variants:
        parts = v.split(":")
        env, lib = {}, None
        for p in parts[1:]:
            k, _, val = p.partition("=")
            if k == "lib":
                lib = val
            else:
                env[k] = val
        variants.append((parts[0], lib, env))
    res = {v[0]: [] for v in variants}
    launches = {}
    for r in range(a.rounds):
        for name, lib, env in variants:
            e = dict(os.environ)
            e.update(env)
            if lib:
                e["DVGR_LIB"] = os.path.join(ROOT, lib)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--min-seconds", "1.5"] + a.bench.split()
            out = subprocess.run(cmd, env=e, capture_output=True, text=True)
            if out.returncode != 0:
                print(out.stderr[-2000:], file=sys.stderr)
                raise SystemExit(f"variant {name} failed")
            d = json.loads(out.stdout.strip().splitlines()[-1])
            if d.get("suspect_setup"):       # bench.py: the reducer trials flagged the stream set-up -- not a measurement to log as one
                raise SystemExit(f"variant {name}: bench.py reports suspect_setup (config.distributed.reducer_trial); refusing to log the row")
            res[name].append(d["ms_per_step"])
            launches[name] = d["config"].get("launches_per_step")
            print(f"round {r} {name}: {d['ms_per_step']:.4f} ms/step", flush=True)
    # rows go to profiles/experiments_r06.tsv AND to gpurun_out/ab_rows/<experiment>.tsv: a GPU box starts from the repo's
    # copy of profiles/, so only gpurun_out/ carries this run's rows back (tools/ab_merge.py folds them in)
    side = os.path.join(ROOT, "gpurun_out", "ab_rows")
    os.makedirs(side, exist_ok=True)
    hdr = "date\texperiment\tvariant\tlib\tenv\tbuild\tmedian_ms\tmin_ms\trounds\tlaunches_per_step\tbench_args\n"
    new = not os.path.exists(a.out)
    with open(a.out, "a") as f, open(os.path.join(side, a.name + ".tsv"), "w") as g:
        if new:
            f.write(hdr)
        for name, lib, env in variants:
            xs = sorted(res[name])
            med = xs[len(xs) // 2]
            row = "\t".join([datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), a.name, name, lib or "libdvgr_hip.so",
                             " ".join(f"{k}={v}" for k, v in env.items()) or "-", bench.build_id(), f"{med:.4f}", f"{xs[0]:.4f}",
                             str(a.rounds), str(launches[name]), a.bench or "-"]) + "\n"
            f.write(row)
            g.write(row)
            print(f"{a.name}\t{name}\tmedian {med:.4f} ms\tmin {xs[0]:.4f} ms")


if __name__ == "__main__":
    main()
