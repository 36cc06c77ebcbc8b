#!/bin/bash
# Round 4, final build: (1) every BASELINE configuration as a bench line, un-profiled; (2) the same command under
# rocprofv3 --kernel-trace --stats; (3) HBM traffic (WRITE_SIZE / FETCH_SIZE in separate PMC passes, scripts/gpu_hbm_all.sh);
# (4) VALU / wave counters (three PMC passes, scripts/gpu_pmc_configs3.sh).  Results under gpurun_out/r04_final/; the summaries to
# keep are copied to profiles/ by scripts/r4_collect.py here in the container.   usage: bash scripts/gpu_r4_final.sh [configs…]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFGS="${*:-c1 c2 c3 c4 c5}"
OUT=$R/gpurun_out/r04_final
mkdir -p $OUT
cd $R
for c in $CFGS; do
  python bench.py --config $c > $OUT/${c}_bench.json 2> $OUT/${c}_bench.err || { tail -20 $OUT/${c}_bench.err; exit 1; }
  echo "bench $c done"
done
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$c -o p --output-format csv -- python3 $R/bench.py --config $c --no-variants --no-cpu-baseline > $OUT/${c}_bench_profiled.json 2> $OUT/${c}_prof.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "profiled bench $c timed out"; exit $rc; fi
  cp $(find $OUT/prof_$c -name '*kernel_stats.csv' | head -1) $OUT/${c}_kernel_stats.csv && rm -rf $OUT/prof_$c
  echo "kernel-trace $c done"
done
CONFIGS="$CFGS" bash $R/scripts/gpu_hbm_all.sh > $OUT/hbm.txt 2>&1 || { tail -5 $OUT/hbm.txt; exit 1; }
echo "hbm passes done"
CONFIGS="$CFGS" PMC_JSON=$OUT/pmc.json bash $R/scripts/gpu_pmc_configs3.sh > $OUT/pmc_table.md 2> $OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
echo "pmc passes done"
# keep the merged-back payload small: raw counter csvs stay on the box
cp $R/gpurun_out/hbm3/*.json $OUT/ 2>/dev/null
rm -rf $R/gpurun_out/hbm3 $R/gpurun_out/pmc3
cat $OUT/pmc_table.md
