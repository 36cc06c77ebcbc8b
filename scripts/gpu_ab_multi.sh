#!/bin/bash
# A/B/C… of several builds of the library on the same box: build/<name>/libraymarcher_amd.so for every name in $BUILDS (the
# in-tree library is saved first and restored at the end), alternating; prints the HIP-event kernel time of the bench frames.
# Usage: BUILDS="base ilp" CFGS="c3 c2" scripts/gpu_ab_multi.sh [rounds]
set -e
R=$GRAFT_REPO_ROOT
cd $R
rounds=${1:-2}
mkdir -p gpurun_out/ab
cp raymarcher_amd/lib/libraymarcher_amd.so /tmp/intree.so
trap 'cp /tmp/intree.so raymarcher_amd/lib/libraymarcher_amd.so' EXIT
for cfg in ${CFGS:-c3}; do
  for i in $(seq 1 $rounds); do
    for which in $BUILDS; do
      cp build/$which/libraymarcher_amd.so raymarcher_amd/lib/libraymarcher_amd.so
      python bench.py --config $cfg --no-variants --no-cpu-baseline 2>gpurun_out/ab/last.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg $which', d['value'], 'Mpixel/s  ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" | tee -a gpurun_out/ab/multi.txt || { echo "$cfg $which FAILED"; tail -3 gpurun_out/ab/last.err; }
    done
  done
done
