#!/bin/bash
# Round-2 session B: PMC passes over the 4K bulb frame (reference formulation) + every BASELINE config timed.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  echo "== pmc pass $i: $set"
  rm -rf gpurun_out/pmc_r2_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_r2_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/pmc_r2_$i.log 2>&1
  rc=$?
  echo "rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_r2_$i.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
echo "== configs"
timeout -k 10 600 python scripts/measure_configs.py gpurun_out/configs_r2.md > gpurun_out/configs_r2.log 2>&1
echo "rc=$?"; tail -15 gpurun_out/configs_r2.log
