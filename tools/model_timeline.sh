set -e
R=$PWD; O=$R/gpurun_out/km2; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $O/trace -o t -- python3 $R/bench.py --model --steps 10 --warmup 5 --min-seconds 0.2 --no-cpu-baseline > $O/trace.log 2>&1
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python tools/timeline.py $DB -1 k_words_fwd > $O/timeline_model.txt
rm -rf $O/trace
