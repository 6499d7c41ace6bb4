#!/bin/bash
# Whole-model step with the native reducer on one rank under different stream set-ups (each a fresh process), against the
# plain step: which set-up is slow, and (AMD_LOG_LEVEL=4) where the host spends the time.  GPU box.  Output: gpurun_out/so/
set -u
OUT=gpurun_out/so; mkdir -p $OUT
export DVGR_DDP_SELFTEST=1 DVGR_BENCH_TRIAL_CHILD=1 MASTER_ADDR=127.0.0.1
run() {   # name, trial kind, extra env...
  local name=$1 kind=$2; shift 2
  env MASTER_PORT=$((29600 + RANDOM % 300)) "$@" python bench.py --model --no-cpu-baseline --trial $kind > $OUT/$name.json 2> $OUT/$name.err
  echo "$name: $(cut -c1-60 $OUT/$name.json)"
}
run plain plain
run native native
run native_noenc native DVGR_NO_ENCODER_OVERLAP=1
run python python
run python_noenc python DVGR_NO_ENCODER_OVERLAP=1
run native_early_ecCs native DVGR_STREAMS_EARLY=1 DVGR_STREAM_ORDER=encoder,copy,comm,side
run native_early_Cesc native DVGR_STREAMS_EARLY=1 DVGR_STREAM_ORDER=comm,encoder,side,copy
run native_early_eCsc native DVGR_STREAMS_EARLY=1 DVGR_STREAM_ORDER=encoder,comm,side,copy
run native_hwq2 native GPU_MAX_HW_QUEUES=2
run native_hwq1 native GPU_MAX_HW_QUEUES=1
run plain_hwq2 plain GPU_MAX_HW_QUEUES=2
run native_log native AMD_LOG_LEVEL=4
python tools/amdlog_hist.py $OUT/native_log.err 0.6 > $OUT/native_log_hist.txt 2>&1
tail -c 3000000 $OUT/native_log.err | gzip > $OUT/native_log_tail.gz; rm -f $OUT/native_log.err
run plain_log plain AMD_LOG_LEVEL=4
python tools/amdlog_hist.py $OUT/plain_log.err 0.6 > $OUT/plain_log_hist.txt 2>&1
tail -c 3000000 $OUT/plain_log.err | gzip > $OUT/plain_log_tail.gz; rm -f $OUT/plain_log.err
head -30 $OUT/native_log_hist.txt
