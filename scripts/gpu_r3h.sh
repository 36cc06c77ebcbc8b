#!/bin/bash
# calibrate SQ_INSTS_VALU against a kernel whose instruction count is known (scripts/microbench/valu_rate2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# built here from its source (the binary is not tracked), with the product's flags
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o $R/scripts/microbench/valu_rate2 $R/scripts/microbench/valu_rate2.hip
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_vr2 -- $R/scripts/microbench/valu_rate2 > $R/gpurun_out/pmc_vr2.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
rows=defaultdict(dict)
for f in glob.glob("$R/gpurun_out/pmc_vr2/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d=rows[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]]=float(r["Counter_Value"]); d["name"]=r["Kernel_Name"][:40]; d["grid"]=int(r["Grid_Size"]); d["dur"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k in sorted(rows)[:60]:
    d=rows[k]
    waves=d["grid"]/64
    print(k, d["name"], "waves", int(waves), "ms %.3f"%(d["dur"]/1e6), "INSTS_VALU/wave %.0f"%(d.get("SQ_INSTS_VALU",0)/waves), "ACTIVE_INST_VALU(quad)/INSTS %.2f"%(d.get("SQ_ACTIVE_INST_VALU",0)/max(d.get("SQ_INSTS_VALU",1),1)), "cycles/instr/SIMD %.2f"%(d["dur"]*1e-9*2.4e9*1024/max(d.get("SQ_INSTS_VALU",1),1)))
PY
grep "x16\|class mix" $R/gpurun_out/pmc_vr2.log | head -20
