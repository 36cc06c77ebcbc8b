#!/bin/bash
# bench.py under alternative environment settings, alternating, fresh process each.  Usage: gpu_env_bench.sh "A=1" "A=2" ...
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/eb_${i}_$rep.log 2>&1 || { echo "FAILED $e"; tail -3 gpurun_out/eb_${i}_$rep.log; exit 1; }
    echo "$e rep $rep: $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/eb_${i}_$rep.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/eb_${i}_$rep.log | head -1)"
  done
done
