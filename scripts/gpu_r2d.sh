#!/bin/bash
# kernel-trace stats of the bench command (no variants: every render_kernel<true,0> launch is a headline-mode frame)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/rocprof.log 2>&1
echo rc=$?
grep '^{"metric"' gpurun_out/rocprof.log | tail -1 | cut -c1-400
