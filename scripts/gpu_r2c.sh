#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python scripts/measure_configs.py gpurun_out/configs_r2_auto.md 2>/dev/null | tail -12
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
