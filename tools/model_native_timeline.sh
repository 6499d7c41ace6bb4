#!/bin/bash
# Kernel timelines of one whole-model step: plain, and with the native reducer on one rank (the slow set-up of the round-4 probe).
set -e
R=$PWD; O=$R/gpurun_out/mnt; mkdir -p $O; export TMPDIR=/tmp
export DVGR_DDP_SELFTEST=1 DVGR_BENCH_TRIAL_CHILD=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29711
cd /tmp
for k in plain native; do
  rocprofv3 --kernel-trace -d $O/trace_$k -o t -- python3 $R/bench.py --model --no-cpu-baseline --trial $k > $O/trace_$k.log 2>&1
  DB=$(find $O/trace_$k -name "*.db" | head -1)
  python3 $R/tools/timeline.py $DB -1 k_words_fwd > $O/timeline_$k.txt
  rm -rf $O/trace_$k
  tail -1 $O/timeline_$k.txt
done
