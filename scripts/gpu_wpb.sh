#!/bin/bash
# Waves per workgroup (RM_WAVES_PER_BLOCK = 1 / 2 / 4) per configuration, fresh process each.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in $1; do
  for w in 1 2 4; do
    RM_WAVES_PER_BLOCK=$w RM_ONLY=$c RM_NO_COUNT=1 timeout -k 10 300 python scripts/measure_configs.py > gpurun_out/wpb_${w}_$c.log 2>&1 || { echo "FAILED $w $c"; tail -3 gpurun_out/wpb_${w}_$c.log; exit 1; }
    echo "wpb $w: $(grep '^| C\|^| sea\|^| area\|^| RC\|^| SKY' gpurun_out/wpb_${w}_$c.log | cut -c1-140)"
  done
done
