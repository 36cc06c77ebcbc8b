#!/bin/bash
# generic (one lane per pixel) vs pipeline, same process settings
mkdir -p gpurun_out
for path in 1 3 4; do
  echo -n "RM_KERNEL_PATH=$path : "
  RM_KERNEL_PATH=$path timeout -k 5 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'])" || exit 1
done
