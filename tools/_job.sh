set -o pipefail
mkdir -p gpurun_out/j19
DVGR_LIB=$PWD/dualvgr-videoqa_amd/libdvgr_hip_x48.so timeout -k 10 600 python -m pytest tests/test_hip_gatb.py tests/test_hip_gatf.py -m gpu -q > gpurun_out/j19/tests.log 2>&1; tail -2 gpurun_out/j19/tests.log
python tools/ab.py xcd_map_4heads_x_half_rows_c2 --rounds 4 --bench "--config c2" heads2_rows16 heads4_rows8:lib=dualvgr-videoqa_amd/libdvgr_hip_x48.so > gpurun_out/j19/ab_c2.log 2>&1 || { tail -20 gpurun_out/j19/ab_c2.log; exit 1; }
tail -2 gpurun_out/j19/ab_c2.log
python tools/ab.py xcd_map_4heads_x_half_rows_c3 --rounds 3 --bench "--config c3" heads2_rows16 heads4_rows8:lib=dualvgr-videoqa_amd/libdvgr_hip_x48.so > gpurun_out/j19/ab_c3.log 2>&1 && tail -2 gpurun_out/j19/ab_c3.log
