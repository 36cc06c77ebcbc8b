#!/bin/bash
# Round-2 session A: parity tests, bench, kernel stats, VALU microbenchmark (EXEC-mask sensitivity).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { # name, timeout, cmd...
  local name=$1 t=$2; shift 2
  echo "== $name" | tee -a gpurun_out/progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/progress.log
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name, stopping"; exit $rc; fi
  return $rc
}
step tests 900 python -m pytest tests -m gpu -q "${PYTEST_ARGS:--x}"
TESTS_RC=$?
step bench 600 python bench.py --steps "${STEPS:-20}" --warmup 3
step microbench 120 scripts/microbench/valu_rate
if [ "${PROFILE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants
  find gpurun_out/prof -name '*kernel_stats*' | head -3
fi
exit $TESTS_RC
