#!/bin/bash
# A/B of the settled tile order (RM_TILE_ORDER_SETTLE=0: re-sort every frame) on c3 / c2 and on the N = 8 shard of c3.
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/settle_ab.txt
: > $OUT
python -m pytest tests -m gpu -x -q -k "tile_order or tile_shape or kernel_path or config" >> $OUT 2>&1
for cfg in c3 c2; do
  for v in 0 3; do
    for rep in 1 2; do
      echo "== $cfg RM_TILE_ORDER_SETTLE=$v rep $rep" >> $OUT
      RM_TILE_ORDER_SETTLE=$v python bench.py --config $cfg --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))" >> $OUT
    done
  done
done
for v in 0 3; do
  echo "== submit_rate_probe RM_TILE_ORDER_SETTLE=$v" >> $OUT
  RM_TILE_ORDER_SETTLE=$v python scripts/submit_rate_probe.py c3 8 >> $OUT 2>&1 || true
done
