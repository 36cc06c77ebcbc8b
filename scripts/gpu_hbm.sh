#!/bin/bash
# HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot limits).
set -u
mkdir -p gpurun_out/hbm
export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  echo "== $c"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/hbm/$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/hbm/$c.log 2>&1
  rc=$?; echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
