#!/bin/bash
# One GPU call's worth of evidence for a build: every bench configuration, rocprofv3 kernel trace (summary + one step's
# timeline) and the three separate PMC passes of the default bench (MI355X_MICROARCH.md, rocprofv3 PMC slots), turned into
# the text / JSON forms kept under profiles/.  (The MFMA-busy pass runs with DVGR_FORKS_BY_RECORD=1: a dispatch that carries a fork's
# stop event reads longer by the signal's path in GRBM_GUI_ACTIVE -- 0.195 instead of 0.217 at C2 -- while its kernel time is the same.)  Output: gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   tools/round_evidence.sh <tag> [notests]
set -o pipefail
T=${1:-ev}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
if [ "$2" != "notests" ]; then python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
for c in c1 c3 c4 c5; do python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2>/dev/null || exit 1; done
python bench.py --model --no-cpu-baseline > $O/bench_model.json 2>/dev/null || exit 1
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_c2_bf16.json 2>/dev/null || exit 1
python bench.py --dtype fp32 --no-cpu-baseline > $O/bench_c2_fp32.json 2>/dev/null || exit 1
DVGR_DDP_SELFTEST=1 python bench.py --no-cpu-baseline > $O/bench_c2_ddp_selftest.json 2>/dev/null || exit 1
DVGR_DDP_SELFTEST=1 python bench.py --reducer python --no-cpu-baseline > $O/bench_c2_ddp_selftest_python.json 2>/dev/null || exit 1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o c2 -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 0.5 --no-cpu-baseline > $O/stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o c2 -- python3 $R/bench.py --steps 4 --warmup 1 --min-seconds 0.2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o c2 -- python3 $R/bench.py --steps 4 --warmup 1 --min-seconds 0.2 --no-cpu-baseline > $O/pmc_write.log 2>&1 || exit 1
DVGR_FORKS_BY_RECORD=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o c2 -- python3 $R/bench.py --steps 4 --warmup 1 --min-seconds 0.2 --no-cpu-baseline > $O/pmc_sq.log 2>&1 || exit 1
cd $R
DB=$(find $O/stats -name "*.db" | head -1)
python tools/rocpd_export.py stats $DB $O/kernel_stats.csv $O/summary.txt || exit 1
python tools/timeline.py $DB > $O/timeline.txt || exit 1
for p in fetch write sq; do
  D=$(find $O/pmc_$p -name "*.db" | head -1)
  python tools/rocpd_export.py counters $D $O/pmc_$p.csv || exit 1
done
python tools/pmc_collect.py $O/c2 c2 bf16x3 auto $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_sq.csv > $O/pmc_collect.log 2>&1 || { tail -5 $O/pmc_collect.log; exit 1; }
cp profiles/pmc_c2_bf16x3.json $O/pmc_c2_bf16x3.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_sq.csv
BENCH_ARGS="--config c5" bash tools/kstats.sh $O c5=dualvgr-videoqa_amd/libdvgr_hip.so || exit 1      # summary_c5.txt, timeline_c5.txt
for f in $O/bench_*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'ms')")"; done
