#!/bin/bash
# Per-kernel statistics (rocprofv3 --kernel-trace) of bench.py for several library builds, one GPU call.
#   tools/kstats.sh <outdir> <name>=<lib.so> [<name>=<lib.so> ...]     (extra bench arguments: BENCH_ARGS="...")
O=$1; shift
R=$PWD; mkdir -p $O; export TMPDIR=/tmp
for V in "$@"; do
  NAME=${V%%=*}; LIB=${V#*=}
  cd /tmp
  DVGR_LIB=$R/$LIB rocprofv3 --kernel-trace -d $O/trace_$NAME -o t -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 0.3 --no-cpu-baseline $BENCH_ARGS > $O/trace_$NAME.log 2>&1 || exit 1
  cd $R
  DB=$(find $O/trace_$NAME -name "*.db" | head -1)
  python tools/rocpd_export.py stats $DB $O/stats_$NAME.csv $O/summary_$NAME.txt || exit 1
  python tools/timeline.py $DB > $O/timeline_$NAME.txt || exit 1
  rm -rf $O/trace_$NAME $O/stats_$NAME.csv
done
