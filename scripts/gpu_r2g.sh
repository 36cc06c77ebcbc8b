#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cfg=${1:-C2}
i=0
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcx_${cfg}_$i
  RM_ONLY=$cfg RM_NO_COUNT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcx_${cfg}_$i -- python scripts/measure_configs.py > gpurun_out/pmcx_${cfg}_$i.log 2>&1
  echo "$cfg pass $i rc=$?"; tail -2 gpurun_out/pmcx_${cfg}_$i.log | cut -c1-300
done
