#!/bin/bash
for f in 1 4 64; do
  echo -n "path 2 flush=$f : "
  RM_KERNEL_PATH=2 RM_PIPE_FLUSH=$f timeout -k 5 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['stage_ms'])" || exit 1
done
