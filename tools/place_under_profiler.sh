set -u
R=$PWD; O=$R/gpurun_out/pl; mkdir -p $O; export TMPDIR=/tmp DVGR_PLACE_LOG=1
cd /tmp
for i in 1 2 3 4 5 6 7 8; do
  rocprofv3 --kernel-trace -d $O/t$i -o t -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 0.3 --no-cpu-baseline > $O/run$i.json 2> $O/run$i.err
  DB=$(find $O/t$i -name "*.db" | head -1)
  python3 $R/tools/timeline.py $DB > $O/timeline$i.txt
  echo "run $i: $(tail -1 $O/timeline$i.txt) queues $(awk '{print $4}' $O/timeline$i.txt | sort | uniq -c | tr '\n' ' ')"
  grep "dvgr place" $O/run$i.err
  rm -rf $O/t$i
done
