#!/bin/bash
# Kernel statistics of the appearance encoder alone (tools/clip_time.py) for one library build.  usage: tools/clip_trace.sh <outdir> [lib.so]
O=$1; R=$PWD; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
DVGR_LIB=${2:+$R/$2} CLIP_STOCK=0 rocprofv3 --kernel-trace -d $O/trace -o t -- python3 $R/tools/clip_time.py > $O/trace.log 2>&1 || exit 1
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python tools/rocpd_export.py stats $DB $O/stats.csv $O/summary.txt || exit 1
rm -rf $O/trace $O/stats.csv
