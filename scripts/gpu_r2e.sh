#!/bin/bash
# PMC passes for the other BASELINE configurations (one configuration per process), then all configs with work counters.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in C1 C1@4K C2 C2@4K C3 "C4 same" "C4 volumetric.json" C5 RC; do
  i=0
  for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
    i=$((i+1))
    rm -rf gpurun_out/pmc_${cfg// /}_$i
    RM_ONLY="$cfg" RM_NO_COUNT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${cfg// /}_$i -- python scripts/measure_configs.py > gpurun_out/pmc_${cfg// /}_$i.log 2>&1
    rc=$?; echo "$cfg pass $i rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
done
timeout -k 10 600 python scripts/measure_configs.py gpurun_out/configs_r2_counts.md 2>&1 | grep -v amdgpu.ids | tail -12
