#!/bin/bash
# Stage split (primary march / surface+normals / shadow marches / shading) of the 4K bulb frame through pipeline B,
# for both evaluation schemes of the Mandelbulb step.
set -u
mkdir -p gpurun_out
for ev in reference algebraic; do
  for path in 1 3; do
    echo "== eval=$ev path=$path" | tee -a gpurun_out/stage.log
    RM_KERNEL_PATH=$path timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --bulb-eval $ev 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'])" | tee -a gpurun_out/stage.log || exit 1
  done
done
