#!/bin/bash
# wave-tile shape experiment: alternative builds of the library are swapped in by path
for tw in 8 4 16; do
  lib=raymarcher_amd/lib/libraymarcher_amd.so
  if [ $tw != 8 ]; then cp raymarcher_amd/lib/libraymarcher_amd_tw$tw.so /tmp/lib_tw.so; cp $lib /tmp/lib_orig.so; cp /tmp/lib_tw.so $lib; fi
  echo -n "tile ${tw}x$((64/tw)) : "
  timeout -k 5 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" || exit 1
  if [ $tw != 8 ]; then cp /tmp/lib_orig.so $lib; fi
done
