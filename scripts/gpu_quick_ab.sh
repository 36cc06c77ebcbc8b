#!/bin/bash
# quick check of a kernel change: the parity tests that touch it, then bench lines of the configurations named in $CFGS
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/quick_ab.txt
: > $OUT
python -m pytest tests -m gpu -x -q ${KEXPR:+-k "$KEXPR"} >> $OUT 2>&1 || { tail -30 $OUT; exit 1; }
tail -2 $OUT
for cfg in ${CFGS:-c5 c2 c1}; do
  for rep in 1 2; do
    python bench.py --config $cfg --steps ${STEPS:-20} --warmup 12 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))" | tee -a $OUT
  done
done
