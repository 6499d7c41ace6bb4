set -o pipefail
mkdir -p gpurun_out/j10
timeout -k 10 600 python -m pytest tests/test_hip_gatb.py -m gpu -q > gpurun_out/j10/tests_gatb.log 2>&1; tail -3 gpurun_out/j10/tests_gatb.log
DVGR_LIB=$PWD/dualvgr-videoqa_amd/libdvgr_hip_clk.so python tools/gatb_clocks.py 64 20 > gpurun_out/j10/clk_gatb_c2.log 2>&1; tail -10 gpurun_out/j10/clk_gatb_c2.log
python tools/ab.py fused_gat_bwd_v4_c2 --rounds 3 --bench "--config c2" two_launches:DVGR_NO_FUSED_GAT_BWD=1 fused > gpurun_out/j10/ab_c2.log 2>&1 || { tail -20 gpurun_out/j10/ab_c2.log; exit 1; }
tail -2 gpurun_out/j10/ab_c2.log
timeout -k 10 600 python -m pytest tests/test_hip_gatf.py -m gpu -q > gpurun_out/j10/tests_gatf.log 2>&1; tail -2 gpurun_out/j10/tests_gatf.log
