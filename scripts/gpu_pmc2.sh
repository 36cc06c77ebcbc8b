#!/bin/bash
# PMC comparison of the bulb schedules (RM_KERNEL_PATH=1,2,3): one rocprofv3 pass per (path, counter set).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_CYCLES"
B="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
C="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_WAVES GRBM_GUI_ACTIVE"
for path in ${PATHS:-1 2 3}; do
  for set in A B C; do
    eval "ctrs=\$$set"
    echo "== path $path set $set"
    RM_KERNEL_PATH=$path timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc/p${path}_$set -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/p${path}_$set.log 2>&1
    rc=$?; echo "rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
done
