#!/bin/bash
# PMC passes over the 4K bulb frame for both evaluation schemes of the step (counters only with --kernel-trace).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for ev in reference algebraic; do
  i=0
  for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
             "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
    i=$((i+1))
    echo "== $ev pass $i: $set"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc3_${ev}_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --bulb-eval $ev > gpurun_out/pmc3_${ev}_$i.log 2>&1
    rc=$?
    echo "rc=$rc"
    if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc3_${ev}_$i.log; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
done
