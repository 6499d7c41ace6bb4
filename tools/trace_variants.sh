#!/bin/bash
# rocprofv3 kernel-trace timelines of one step for several DVGR_SCHED values of the aids build (one GPU call).
#   tools/trace_variants.sh <outdir> <sched> [<sched> ...]
O=$1; shift
R=$PWD; mkdir -p $O; export TMPDIR=/tmp
for S in "$@"; do
  cd /tmp
  DVGR_LIB=$R/dualvgr-videoqa_amd/libdvgr_hip_aids.so DVGR_SCHED=$S rocprofv3 --kernel-trace -d $O/trace_s$S -o t -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 0.1 --no-cpu-baseline > $O/trace_s$S.log 2>&1 || exit 1
  cd $R
  DB=$(find $O/trace_s$S -name "*.db" | head -1)
  python tools/timeline.py $DB > $O/timeline_s$S.txt || exit 1
  rm -rf $O/trace_s$S
done
