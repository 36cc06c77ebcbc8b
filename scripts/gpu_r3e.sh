#!/bin/bash
# chunk-size sweep of the wavefront pipeline (RC at 4K, C5 at 8K): "slot ray-floor pixel-floor max-chunk flush"
export RM_NO_COUNT=1 RM_ONLY=${RM_ONLY:-RC,C5} RM_KERNEL_PATH=5
for cfg in ${RM_SWEEP:-"256,64,64,64,16" "256,256,256,256,16" "256,512,512,512,16" "256,1024,1024,1024,16" "256,64,64,2048,16" "256,256,256,2048,16" "256,256,256,2048,8" "1024,1024,1024,4096,16"}; do
  IFS=, read a b c m d <<< "$cfg"
  echo "== slot $a ray $b pixel $c max $m flush $d"
  RM_WF_SLOT_CHUNK=$a RM_WF_RAY_CHUNK=$b RM_WF_PIXEL_CHUNK=$c RM_WF_MAX_CHUNK=$m RM_WF_FLUSH=$d python scripts/measure_configs.py 2>/dev/null | cut -c1-50,70-120
done
