#!/bin/bash
# Time the configurations of scripts/measure_configs.py with alternative builds of the library (raymarcher_amd/lib/exp/*.so,
# built by hand with `make OUT=../lib/exp/<name>.so EXTRA=-D...`): each in a fresh process, the variant copied over the
# library path on the GPU box's scratch copy of the repo.  Usage: bash scripts/gpu_variants.sh "C1 C2 C5" w5 w6
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cfgs=$1; shift
cp raymarcher_amd/lib/libraymarcher_amd.so /tmp/lib_base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/lib_base.so raymarcher_amd/lib/libraymarcher_amd.so; else cp raymarcher_amd/lib/exp/lib_$v.so raymarcher_amd/lib/libraymarcher_amd.so; fi
  for c in $cfgs; do
    RM_ONLY=$c RM_NO_COUNT=1 timeout -k 10 300 python scripts/measure_configs.py > gpurun_out/var_${v}_$c.log 2>&1 || { echo "FAILED $v $c"; tail -3 gpurun_out/var_${v}_$c.log; exit 1; }
    echo "$v: $(grep '^| C\|^| sea\|^| area\|^| RC\|^| SKY' gpurun_out/var_${v}_$c.log | cut -c1-160)"
  done
done
cp /tmp/lib_base.so raymarcher_amd/lib/libraymarcher_amd.so
