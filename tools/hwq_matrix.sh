for Q in "" 8 16; do
  if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
  echo "== GPU_MAX_HW_QUEUES=$Q"
  python bench.py --no-cpu-baseline --min-seconds 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unit plain        ', round(d['ms_per_step'],4))"
  DVGR_DDP_SELFTEST=1 python bench.py --no-cpu-baseline --min-seconds 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unit ddp native   ', round(d['ms_per_step'],4))"
  DVGR_DDP_SELFTEST=1 python bench.py --reducer python --no-cpu-baseline --min-seconds 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unit ddp python   ', round(d['ms_per_step'],4))"
  python bench.py --model --no-cpu-baseline --min-seconds 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('model plain       ', round(d['ms_per_step'],4))"
  DVGR_DDP_SELFTEST=1 python bench.py --model --no-cpu-baseline --min-seconds 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('model ddp python  ', round(d['ms_per_step'],4))"
done
