#!/bin/bash
# wavefront pipeline: parity tests, then A/B timings against the one-lane-per-pixel kernel
set -e
python -m pytest tests/test_gpu_wavefront.py -x -q > gpurun_out/r03_b_wf_pytest.log 2>&1 || { tail -30 gpurun_out/r03_b_wf_pytest.log; exit 1; }
tail -2 gpurun_out/r03_b_wf_pytest.log
export RM_NO_COUNT=1 RM_ONLY=C2,C2@4K,RC,C5
RM_KERNEL_PATH=1 python scripts/measure_configs.py gpurun_out/r03_b_mono.md > /dev/null
RM_KERNEL_PATH=5 python scripts/measure_configs.py gpurun_out/r03_b_wf.md > /dev/null
for T in 8 32; do RM_WF_FLUSH=$T RM_KERNEL_PATH=5 RM_ONLY=RC,C5 python scripts/measure_configs.py gpurun_out/r03_b_wf_T$T.md > /dev/null; done
RM_WF_WAVES_PER_CU=16 RM_KERNEL_PATH=5 RM_ONLY=RC,C5 python scripts/measure_configs.py gpurun_out/r03_b_wf_w16.md > /dev/null
tail -n +3 gpurun_out/r03_b_mono.md gpurun_out/r03_b_wf.md gpurun_out/r03_b_wf_T8.md gpurun_out/r03_b_wf_T32.md gpurun_out/r03_b_wf_w16.md
