#!/bin/bash
# PMC of the headline frame (one configuration per process): instruction counts, lane use, resident waves.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcc3_$i
  RM_ONLY=C3 RM_NO_COUNT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcc3_$i -- python scripts/measure_configs.py > gpurun_out/pmcc3_$i.log 2>&1
  echo "pass $i rc=$?"; grep "^| C3" gpurun_out/pmcc3_$i.log | cut -c1-120
done
