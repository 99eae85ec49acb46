#!/bin/bash
# GPU box: matrix-pipe utilisation per kernel from PMC counters (own pass, no tracing domains besides the kernel trace).
#   utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_m -- python3 $R/bench.py --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, collections
f=glob.glob("/tmp/pmc_m/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="GRBM_GUI_ACTIVE": cnt[k]+=1
out={}
for k,v in agg.items():
    if v.get("SQ_INSTS_MFMA",0)<=0: continue
    util=v["SQ_VALU_MFMA_BUSY_CYCLES"]/(v["GRBM_GUI_ACTIVE"]/8.0*1024.0)
    out[k]={"launches":cnt[k],"mfma_busy_cycles_per_launch":v["SQ_VALU_MFMA_BUSY_CYCLES"]/cnt[k],"gui_active_per_launch":v["GRBM_GUI_ACTIVE"]/cnt[k],
            "mfma_insts_per_launch":v["SQ_INSTS_MFMA"]/cnt[k],"mfma_pipe_utilisation":util}
json.dump({"command":"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES -- python3 bench.py --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline",
           "note":"utilisation = MFMA busy cycles summed over SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); B=64 VG forwards","kernels":out},
          open("$R/gpurun_out/pmc_mfma_util.json","w"),indent=1)
for k,v in sorted(out.items(), key=lambda kv:-kv[1]["mfma_busy_cycles_per_launch"]*kv[1]["launches"]):
    print("%-60s launches %5d  MFMA pipe utilisation %.3f"%(k[:60], v["launches"], v["mfma_pipe_utilisation"]))
PY
