# Timeline of the one-rank data-parallel self-test under rocprofv3.  The reducer is named explicitly: `--reducer auto` would
# start bench.py's trial children, which a profiler must not trace (they are started without the profiler's preload and tool
# variables anyway: bench.py child_env()).   tools/ddp_timeline.sh [native|inorder|python]
RED=${1:-native}
R=$PWD; O=$R/gpurun_out/ddptl; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
DVGR_DDP_SELFTEST=1 rocprofv3 --kernel-trace -d $O/trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 0.3 --no-cpu-baseline --reducer $RED > $O/trace.log 2>&1 || exit 1
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python tools/timeline.py $DB > $O/timeline.txt
python tools/rocpd_export.py stats $DB $O/stats.csv $O/summary.txt
rm -rf $O/trace $O/stats.csv
grep -iv "dvgr::\|rocclr" $O/summary.txt | head -8 | cut -c1-150
grep -n "q[4-9]\|nccl\|rccl\|Kernel" $O/timeline.txt | head -20 | cut -c1-140
tail -1 $O/timeline.txt
