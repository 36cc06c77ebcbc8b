#!/bin/bash
# PMC passes for one configuration of scripts/measure_configs.py (RM_ONLY=<cfg>): wait / scalar-cache / issue counters.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cfg=${1:-C2}
i=0
rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1
for set in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcy_${cfg}_$i
  RM_ONLY=$cfg RM_NO_COUNT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcy_${cfg}_$i -- python scripts/measure_configs.py > gpurun_out/pmcy_${cfg}_$i.log 2>&1
  echo "$cfg pass $i rc=$?"; tail -2 gpurun_out/pmcy_${cfg}_$i.log | cut -c1-300
done
