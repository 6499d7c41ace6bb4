run() { python bench.py --no-cpu-baseline "${@:2}" 2>gpurun_out/_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['value'],1))"; }
mkdir -p gpurun_out
run plain
DVGR_DDP_SELFTEST=1 run ddp_selftest
run c1 --config c1
DVGR_NO_QUERY_STREAM=1 run c1_noqs --config c1
python -m pytest tests/test_hip_parallel.py tests/test_cabi.py -q 2>&1 | tail -2
