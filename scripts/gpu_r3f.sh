#!/bin/bash
# per-kernel times of the wavefront pipeline on C5 under two chunk settings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "256,256,256,256,16" "256,1024,1024,1024,16" "256,128,128,128,16"; do
  IFS=, read a b c m d <<< "$cfg"
  echo "== slot $a ray $b pixel $c max $m flush $d"
  RM_WF_SLOT_CHUNK=$a RM_WF_RAY_CHUNK=$b RM_WF_PIXEL_CHUNK=$c RM_WF_MAX_CHUNK=$m RM_WF_FLUSH=$d RM_NO_COUNT=1 RM_ONLY=C5 RM_KERNEL_PATH=5 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f_$b -o wf --output-format csv -- python3 $R/scripts/measure_configs.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_f_$b -name '*kernel_stats.csv' | head -1)
  cut -d, -f1-4 $f | sed 's/(rm::SceneBlock[^"]*//' | head -6
done
