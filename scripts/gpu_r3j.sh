#!/bin/bash
# lockstep shadow rays (RM_LOCKSTEP): parity of the table-walk classes, then table-walk frames with and without
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wavefront.py tests/test_scenefile_pixels.py -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1 || { tail -30 gpurun_out/r3j_tests.log; exit 1; }
tail -2 gpurun_out/r3j_tests.log
for on in 1 0; do
  echo "RM_LOCKSTEP=$on"
  RM_LOCKSTEP=$on RM_NO_COUNT=1 RM_ONLY="C2,C2@4K,RC,RC@1080p,RC2,C1@4K" timeout -k 10 300 python scripts/measure_configs.py gpurun_out/r3j_cfg_$on.md > gpurun_out/r3j_cfg_$on.log 2>&1
  cat gpurun_out/r3j_cfg_$on.md
done
