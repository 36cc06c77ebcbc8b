#!/bin/bash
# per-kernel times of the wavefront pipeline on C5 and RC (rocprofv3 kernel trace)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in C5 RC; do
  RM_NO_COUNT=1 RM_ONLY=$c RM_KERNEL_PATH=5 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_wf_$c -o wf_$c --output-format csv -- python3 $R/scripts/measure_configs.py > $R/gpurun_out/r03_c_prof_$c.log 2>&1
  f=$(find $R/gpurun_out/prof_wf_$c -name '*kernel_stats.csv' | head -1)
  cp $f $R/gpurun_out/r03_c_wf_${c}_kernel_stats.csv
  cat $f
done
