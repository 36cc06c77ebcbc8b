#!/bin/bash
# full GPU suite, then A/B timings of the generic classes: one lane per pixel (path 1) vs wavefront (path 5)
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r03_d_pytest.log 2>&1 || { tail -30 gpurun_out/r03_d_pytest.log; exit 1; }
tail -2 gpurun_out/r03_d_pytest.log
export RM_NO_COUNT=1 RM_ONLY=${RM_ONLY:-C2,C2@4K,RC,C5}
RM_KERNEL_PATH=1 python scripts/measure_configs.py gpurun_out/r03_d_mono.md > /dev/null
RM_KERNEL_PATH=5 python scripts/measure_configs.py gpurun_out/r03_d_wf.md > /dev/null
tail -n +3 gpurun_out/r03_d_mono.md gpurun_out/r03_d_wf.md
