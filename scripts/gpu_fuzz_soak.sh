#!/bin/bash
# The soak of the table walk's exactness evidence: the wide random table-walk test with 1000 cases for each
# of three seeds, plus the older generators with 400.  One pytest process per seed, one after the other; writes
# gpurun_out/fuzz_soak.txt (copy to profiles/ by hand).   usage (on the GPU box): bash scripts/gpu_fuzz_soak.sh
set -e -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/fuzz_soak.txt
: > $OUT
CASES=${CASES:-1000}
for SEED in ${SEEDS:-20261012 7 987654321}; do
  echo "== seed $SEED: test_random_tablewalk_scenes_bit_exact, $CASES cases" | tee -a $OUT
  RM_FUZZ_CASES=$CASES RM_FUZZ_SEED=$SEED timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x \
     -k "test_random_tablewalk_scenes_bit_exact" 2>&1 | grep -E "FUZZ_SUMMARY|passed|failed|Error|assert" | tee -a $OUT
done
for SEED in ${OLD_SEEDS:-424242}; do
  echo "== seed $SEED: the older generators (general, all-primitive, bulb class, wavefront class), ${OLD_CASES:-400} cases each" | tee -a $OUT
  RM_FUZZ_CASES=${OLD_CASES:-400} RM_FUZZ_SEED=$SEED timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wavefront.py -m gpu -q -x \
     -k "test_random_scenes_bit_exact or test_random_primitive_scenes_bit_exact or test_random_bulb or test_wavefront_random_scenes_bit_exact" 2>&1 | tail -3 | tee -a $OUT
done
