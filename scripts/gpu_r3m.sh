#!/bin/bash
# scene bounding box on top of the bounding ball (RM_CULL_BOX): parity suite, then the table-walk frames with and without
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3m_tests.log 2>&1 || { tail -30 gpurun_out/r3m_tests.log; exit 1; }
tail -2 gpurun_out/r3m_tests.log
for on in 1 0; do
  echo "RM_CULL_BOX=$on"
  RM_CULL_BOX=$on RM_ONLY="C1,C2,C2@4K,RC,RC@1080p,C1@4K,C5,C5@4K,SKY" timeout -k 10 500 python scripts/measure_configs.py gpurun_out/r3m_cfg_$on.md > gpurun_out/r3m_cfg_$on.log 2>&1
  cat gpurun_out/r3m_cfg_$on.md
done
