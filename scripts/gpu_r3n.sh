#!/bin/bash
# fuzz soak on the new bounds (tight ball, box) and the lockstep class, then the bench lines
set -e
mkdir -p gpurun_out
RM_FUZZ_CASES=1500 RM_FUZZ_SEED=424242 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wavefront.py tests/test_gpu_lockstep.py -m gpu -q -x -k "random" > gpurun_out/r3n_fuzz.log 2>&1 || { tail -30 gpurun_out/r3n_fuzz.log; exit 1; }
tail -1 gpurun_out/r3n_fuzz.log
for c in c1 c2 c3 c4 c5; do
  timeout -k 10 500 python bench.py --config $c > gpurun_out/r3n_$c.json 2> gpurun_out/r3n_$c.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3n_$c.json").read().strip().splitlines()[-1])
print("$c", d["value"], "ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "exec", d["roofline"].get("executed",{}).get("frac"), {k:v.get("ms_per_step") for k,v in d["variants"].items()}, d["parity_check"]["mismatched_words"], d["cpu_baseline"]["value"])
PY
done
