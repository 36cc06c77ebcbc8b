#!/bin/bash
# bench.py under alternative builds of the library (see gpu_variants.sh), alternating, fresh process each.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cp raymarcher_amd/lib/libraymarcher_amd.so /tmp/lib_base.so
for rep in 1 2 3; do
  for v in base "$@"; do
    if [ "$v" = base ]; then cp /tmp/lib_base.so raymarcher_amd/lib/libraymarcher_amd.so; else cp raymarcher_amd/lib/exp/lib_$v.so raymarcher_amd/lib/libraymarcher_amd.so; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/vb_${v}_$rep.log 2>&1 || { echo "FAILED $v"; tail -3 gpurun_out/vb_${v}_$rep.log; exit 1; }
    echo "$v rep $rep: $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/vb_${v}_$rep.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/vb_${v}_$rep.log | head -1)"
  done
done
cp /tmp/lib_base.so raymarcher_amd/lib/libraymarcher_amd.so
