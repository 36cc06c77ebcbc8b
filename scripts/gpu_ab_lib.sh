#!/bin/bash
# A/B of two builds of the library on the same box: build/base/libraymarcher_amd.so (the build before a change) against the
# in-tree one, alternating; prints the HIP-event kernel time of the bench frame.  Usage: scripts/gpu_ab_lib.sh [config] [rounds]
set -e
R=$GRAFT_REPO_ROOT
cd $R
cfg=${1:-c3}; rounds=${2:-3}
mkdir -p gpurun_out/ab
cp raymarcher_amd/lib/libraymarcher_amd.so gpurun_out/ab/new.so.keep
trap 'cp gpurun_out/ab/new.so.keep raymarcher_amd/lib/libraymarcher_amd.so' EXIT  # an interrupted run must not leave the base build in the tree
for i in $(seq 1 $rounds); do
  for which in base new; do
    if [ $which = base ]; then cp build/base/libraymarcher_amd.so raymarcher_amd/lib/libraymarcher_amd.so; else cp gpurun_out/ab/new.so.keep raymarcher_amd/lib/libraymarcher_amd.so; fi
    python bench.py --config $cfg --no-variants --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg $which', d['value'], 'Mpixel/s  ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" | tee -a gpurun_out/ab/${cfg}.txt
  done
done
