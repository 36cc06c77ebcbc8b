#!/bin/bash
# PMC passes over the wavefront pipeline on C5 (and the one-lane-per-pixel kernel beside it): usage gpu_r3c.sh <tag> <path> [config]
set -u
tag=${1:-wf}; path=${2:-5}; cfg=${3:-C5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  RM_NO_COUNT=1 RM_ONLY=$cfg RM_KERNEL_PATH=$path timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/scripts/measure_configs.py > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i rc=$rc"; tail -5 $R/gpurun_out/pmc_${tag}_$i.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R
for k in "wf_march_kernel<0>" "wf_march_kernel<1>" "wf_march_kernel<2>" "wf_surface" "wf_light" "render_kernel"; do
  python3 scripts/pmc_summary.py "$k" gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 gpurun_out/pmc_${tag}_4 2>/dev/null
done > gpurun_out/r03_pmc_${tag}.md
cat gpurun_out/r03_pmc_${tag}.md
