#!/bin/bash
# Round-end evidence: tests, the default bench line, and the same bench under rocprofv3 --kernel-trace --stats.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/final_tests.log
timeout -k 10 600 python bench.py > gpurun_out/final_bench.log 2>&1; echo "bench rc=$?"
rm -rf gpurun_out/final_prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -- python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/final_bench_profiled.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/final_prof -name '*kernel_stats*'
