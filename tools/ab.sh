#!/bin/bash
# A/B of two builds of the library on ONE device (devices differ by several per cent): alternates `bench.py` runs.
#   tools/ab.sh <lib_a.so> <lib_b.so> [bench args...]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in "$A" "$B"; do
    DVGR_LIB=$PWD/$L python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],4), round(d['value'],1))" || exit 1
  done
done
