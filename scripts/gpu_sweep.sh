#!/bin/bash
# A/B sweep of pipeline launch knobs; each run is a separate short bench (stage timings in the JSON line).
mkdir -p gpurun_out
for b in 2 3 4 6 8; do for f in 8 16 32; do
  echo -n "blocks/CU=$b flush=$f : "
  RM_PIPE_BLOCKS_PER_CU=$b RM_PIPE_FLUSH=$f timeout -k 5 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['stage_ms'])" || exit 1
done; done
