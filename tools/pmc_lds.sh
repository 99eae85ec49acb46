#!/bin/bash
# GPU box: LDS activity and matrix-pipe utilisation per kernel of a bench.py workload from PMC counters (two own passes, kernel trace only).
#   lds_active  = SQ_LDS_IDX_ACTIVE  / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs)   -- share of the kernel's time the LDS arrays are busy (per CU)
#   lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                    -- share of those cycles that are bank-conflict replays
#   mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
# PMC_BENCH_ARGS selects the workload (default: configs[4]'s per-GPU share in the bf16 mode); TAG names the output.
R=$GRAFT_REPO_ROOT
ARGS=${PMC_BENCH_ARGS:---config coco --batch 512 --precision bf16}
TAG=${TAG:-coco_bf16}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_l1 /tmp/pmc_l2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/pmc_l1 -- python3 $R/bench.py $ARGS --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d /tmp/pmc_l2 -- python3 $R/bench.py $ARGS --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, collections
def load(d):
    f=glob.glob(d+"/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": cnt[k]+=1
    return agg,cnt
a1,c1=load("/tmp/pmc_l1"); a2,c2=load("/tmp/pmc_l2")
out={}
for k,v in a1.items():
    if v.get("SQ_INSTS_LDS",0)<=0 or k not in a2: continue
    g=v["GRBM_GUI_ACTIVE"]/8.0
    w=a2[k]; g2=w["GRBM_GUI_ACTIVE"]/8.0
    out[k]={"launches":c1[k],"gui_active_cycles_per_launch":g/c1[k],
            "lds_active":v["SQ_LDS_IDX_ACTIVE"]/(g*256.0),"lds_conflict_share":v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_LDS_IDX_ACTIVE"],1.0),
            "mfma_util":w["SQ_VALU_MFMA_BUSY_CYCLES"]/(g2*1024.0),"valu_insts_per_mfma":w["SQ_INSTS_VALU"]/max(w["SQ_INSTS_MFMA"],1.0)}
json.dump({"command":"rocprofv3 --pmc {SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU} -- python3 bench.py $ARGS --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline",
           "kernels":out}, open("$R/gpurun_out/pmc_lds_$TAG.json","w"), indent=1)
for k,v in sorted(out.items(), key=lambda kv:-kv[1]["gui_active_cycles_per_launch"]*kv[1]["launches"]):
    print("%-52s launches %5d  LDS active %.3f  conflict share %.3f  MFMA util %.3f  VALU/MFMA %.1f"%(k[:52], v["launches"], v["lds_active"], v["lds_conflict_share"], v["mfma_util"], v["valu_insts_per_mfma"]))
PY
