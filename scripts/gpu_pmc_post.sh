#!/bin/bash
# PMC passes over the post passes (scripts/measure_post.py): what the FXAA kernel spends its time on.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_post
i=0
for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_post/p$i -- python3 $R/scripts/measure_post.py > $R/gpurun_out/pmc_post/p$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i rc=$rc"; tail -3 $R/gpurun_out/pmc_post/p$i.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R
python3 - <<'PY'
import csv, glob
from collections import defaultdict
vals=defaultdict(lambda: defaultdict(list)); durs=defaultdict(list)
for f in glob.glob("gpurun_out/pmc_post/p*/**/*_counter_collection.csv", recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "anonymous" not in n: continue
        k=n.split("(anonymous namespace)::")[1].split("(")[0]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"])); durs[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in vals:
    a={c: sum(v)/len(v) for c,v in vals[k].items()}
    ms=sum(durs[k])/len(durs[k])/1e6
    print(k, f"{ms:.3f} ms", {c: f"{v:.3g}" for c,v in a.items()})
PY
