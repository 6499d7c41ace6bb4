set -u
OUT=gpurun_out/ns; mkdir -p $OUT
python tools/ab.py caller_stream_c1 --rounds 3 --bench "--config c1 --no-cpu-baseline" null own:DVGR_BENCH_STREAM=1 own_single:DVGR_BENCH_STREAM=1:DVGR_SINGLE_STREAM=1 --out gpurun_out/ab_rows/caller_stream_c1.tsv
python tools/ab.py caller_stream_c2 --rounds 3 --bench "--config c2 --no-cpu-baseline" null own:DVGR_BENCH_STREAM=1 --out gpurun_out/ab_rows/caller_stream_c2.tsv
python tools/ab.py caller_stream_model --rounds 2 --bench "--model --no-cpu-baseline" null own:DVGR_BENCH_STREAM=1 --out gpurun_out/ab_rows/caller_stream_model.tsv
export DVGR_DDP_SELFTEST=1 DVGR_BENCH_TRIAL_CHILD=1 MASTER_ADDR=127.0.0.1
for k in plain native python; do
  MASTER_PORT=$((29600 + RANDOM % 300)) DVGR_BENCH_STREAM=1 python bench.py --model --no-cpu-baseline --trial $k > $OUT/model_$k.json 2> $OUT/model_$k.err; echo "model own-stream $k: $(cut -c1-60 $OUT/model_$k.json)"
  MASTER_PORT=$((29600 + RANDOM % 300)) DVGR_BENCH_STREAM=1 python bench.py --no-cpu-baseline --trial $k > $OUT/unit_$k.json 2> $OUT/unit_$k.err; echo "unit own-stream $k: $(cut -c1-60 $OUT/unit_$k.json)"
done
