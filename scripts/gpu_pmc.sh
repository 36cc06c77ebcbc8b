#!/bin/bash
# PMC pass(es) over a short bench run.  Counters only with --kernel-trace (no sys/hip/hsa trace domains).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${LIST:-0}" = "1" ]; then rocprofv3 -L > gpurun_out/counters_list.txt 2>&1; fi
i=0
for set in "$@"; do
  i=$((i+1))
  echo "== pmc pass $i: $set"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc$i.log 2>&1
  rc=$?
  echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
