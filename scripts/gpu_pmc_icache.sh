#!/bin/bash
# Instruction-fetch counters of the bench frames (round 3): are the march kernels (render_kernel<true,0,0,0> is 227 KB of code,
# the instruction cache 64 KB per CU pair) waiting for instructions?  One rocprofv3 --pmc pass per counter set and configuration.
set -u
mkdir -p gpurun_out/icache
export TMPDIR=/tmp
for cfg in ${CFGS:-c3 c2 c5}; do
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAIT_ANY"; do
    i=$((i+1))
    rm -rf gpurun_out/icache/${cfg}_$i
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/icache/${cfg}_$i -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/icache/${cfg}_$i.log 2>&1
    rc=$?
    echo "$cfg pass $i rc=$rc"
    if [ $rc -ne 0 ]; then tail -5 gpurun_out/icache/${cfg}_$i.log; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
  for k in render_kernel wf_march_kernel; do
    python3 scripts/pmc_summary.py $k gpurun_out/icache/${cfg}_1 gpurun_out/icache/${cfg}_2 gpurun_out/icache/${cfg}_3 2>/dev/null | tee -a gpurun_out/icache/${cfg}_summary.md
  done
  rm -rf gpurun_out/icache/${cfg}_[123]
done
