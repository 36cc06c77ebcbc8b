#!/bin/bash
# PMC passes over every bench configuration (production kernels; per launch averages).  PMC_JSON=<file>: also write, per
# configuration, the frame's aggregate (all production kernels of a frame together) as JSON for bench.py's roofline.valu_issue.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc3
for cfg in ${CONFIGS:-c1 c2 c3 c4 c5}; do
  i=0
  for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc3/${cfg}_$i -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/pmc3/${cfg}_$i.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "$cfg pass $i rc=$rc"; tail -3 $R/gpurun_out/pmc3/${cfg}_$i.log; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
done
cd $R
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
def production(n):
    if "wf_" in n: return True
    m = re.search(r"render_kernel<(\w+), (\d+)", n)
    return bool(m) and m.group(2) == "0"
import json, os
agg={}
print("| configuration | kernel | launches | ms | SQ_INSTS_VALU | lanes live | cycles per VALU instr per SIMD | resident waves / SIMD | s_waitcnt share (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | issue-stall share (SQ_WAIT_INST_ANY) | SALU / VALU |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for cfg in ["c1","c2","c3","c4","c5"]:
    vals=defaultdict(lambda: defaultdict(list)); durs=defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc3/{cfg}_*/**/*_counter_collection.csv", recursive=True):
        seen=set()
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if not production(n): continue
            k=n.split("(")[0].replace("void ","")
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (f, r["Dispatch_Id"]) not in seen:
                seen.add((f, r["Dispatch_Id"])); durs[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in sorted(vals):
        a={c: sum(v)/len(v) for c,v in vals[k].items()}
        ms=sum(durs[k])/len(durs[k])/1e6
        iv=a.get("SQ_INSTS_VALU",0)
        if iv < 1e5: continue
        cyc=ms*1e-3*2.39e9*1024
        g=agg.setdefault(cfg,{"slots":0.0,"lane_instr":0.0,"instr":0.0,"cyc":0.0,"wave_cyc":0.0})
        nl=len(durs[k])/3.0  # launches per pass
        g["instr"]+=iv*nl; g["lane_instr"]+=a.get("SQ_THREAD_CYCLES_VALU",0)*nl; g["cyc"]+=cyc*nl; g["wave_cyc"]+=4*a.get("SQ_WAVE_CYCLES",0)*nl
        print(f"| {cfg} | `{k}` | {len(durs[k])//3} | {ms:.3f} | {iv:.3g} | {a.get('SQ_THREAD_CYCLES_VALU',0)/(64*iv):.0%} | {cyc/iv:.2f} | {4*a.get('SQ_WAVE_CYCLES',0)/cyc:.2f} | {a.get('SQ_WAIT_ANY',0)/max(a.get('SQ_WAVE_CYCLES',1),1):.0%} | {a.get('SQ_WAIT_INST_ANY',0)/max(a.get('SQ_WAVE_CYCLES',1),1):.0%} | {a.get('SQ_INSTS_SALU',0)/iv:.2f} |")
out={}
for cfg,g in agg.items():
    if g["instr"]<=0: continue
    # issue slots: one wave64 VALU instruction per 2 cycles per SIMD at full rate (SIMD-32), every lane live
    out[cfg]={"valu_issue_frac": round(2.0*g["lane_instr"]/64.0/g["cyc"],4), "lanes_live": round(g["lane_instr"]/(64.0*g["instr"]),3),
              "cycles_per_valu": round(g["cyc"]/g["instr"],2), "resident_waves": round(g["wave_cyc"]/g["cyc"],2)}
if os.environ.get("PMC_JSON"):
    json.dump(out, open(os.environ["PMC_JSON"],"w"), indent=1)
PY
