#!/bin/bash
# full GPU suite, every measure_configs line
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3q_tests.log 2>&1 || { tail -30 gpurun_out/r3q_tests.log; exit 1; }
tail -2 gpurun_out/r3q_tests.log
timeout -k 10 900 python scripts/measure_configs.py gpurun_out/r3q_cfg.md > gpurun_out/r3q_cfg.log 2>&1
cat gpurun_out/r3q_cfg.md
