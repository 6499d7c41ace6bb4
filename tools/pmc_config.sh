#!/bin/bash
# The three separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of bench.py for ONE configuration, turned into
# profiles-style records: <outdir>/<config>_pmc_hbm.txt and profiles/pmc_<config>_bf16x3.json (copied to <outdir> as well).
#   tools/pmc_config.sh <outdir> <config>
set -o pipefail
O=$1; C=$2; R=$PWD; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $R/bench.py --config $C --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $R/bench.py --config $C --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python3 $R/bench.py --config $C --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline > $O/pmc_sq.log 2>&1 || exit 1
cd $R
for p in fetch write sq; do
  D=$(find $O/pmc_$p -name "*.db" | head -1)
  python tools/rocpd_export.py counters $D $O/pmc_$p.csv || exit 1
done
python tools/pmc_collect.py $O/$C $C bf16x3 auto $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_sq.csv > $O/pmc_collect.log 2>&1 || { tail -5 $O/pmc_collect.log; exit 1; }
cp profiles/pmc_${C}_bf16x3.json $O/
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_sq.csv
tail -3 $O/${C}_pmc_hbm.txt
