#!/bin/bash
# HBM traffic of every bench configuration: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot limits).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hbm3
for cfg in ${CONFIGS:-c1 c2 c3 c4 c5}; do
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/hbm3/$cfg/$c -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/hbm3/${cfg}_$c.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "$cfg $c rc=$rc"; tail -3 $R/gpurun_out/hbm3/${cfg}_$c.log; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
  python3 $R/scripts/hbm_summary.py $cfg $R/gpurun_out/hbm3/$cfg | tee $R/gpurun_out/hbm3/$cfg.json
done
