#!/bin/bash
# object skip test in the table walk: parity suite, then the table-walk frames
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3o_tests.log 2>&1 || { tail -30 gpurun_out/r3o_tests.log; exit 1; }
tail -2 gpurun_out/r3o_tests.log
RM_NO_COUNT=1 RM_ONLY="C1,C2,C2@4K,RC,RC@1080p,C1@4K,SKY,area" timeout -k 10 500 python scripts/measure_configs.py gpurun_out/r3o_cfg.md > gpurun_out/r3o_cfg.log 2>&1
cat gpurun_out/r3o_cfg.md
