#!/bin/bash
# A/B/C… of several builds of the library on the same box over scripts/measure_configs.py cases (HIP-event kernel ms, 10 frames
# each): build/<name>/libraymarcher_amd.so for every name in $BUILDS ("intree" = the in-tree library), alternating; the in-tree
# library is saved first and restored on exit, whatever happens.
# Usage: BUILDS="base intree" CASES="C2,C2@4K,RC" scripts/gpu_ab_configs.sh [rounds] [out.txt]
set -e
cd ${GRAFT_REPO_ROOT:-.}
rounds=${1:-2}; out=${2:-gpurun_out/ab_configs.txt}
mkdir -p gpurun_out
cp raymarcher_amd/lib/libraymarcher_amd.so /tmp/intree.so
trap 'cp /tmp/intree.so raymarcher_amd/lib/libraymarcher_amd.so' EXIT
: > $out
for i in $(seq 1 $rounds); do
  for which in $BUILDS; do
    if [ $which = intree ]; then cp /tmp/intree.so raymarcher_amd/lib/libraymarcher_amd.so; else cp build/$which/libraymarcher_amd.so raymarcher_amd/lib/libraymarcher_amd.so; fi
    echo "== round $i build $which" | tee -a $out
    RM_ONLY="${CASES:-C2}" RM_NO_COUNT=1 timeout -k 10 300 python scripts/measure_configs.py 2>/dev/null | grep "^| [A-Za-z0-9]" | grep -v "configuration" | cut -d'|' -f2,3 | tee -a $out
  done
done
