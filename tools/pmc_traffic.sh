#!/bin/bash
# GPU box: HBM traffic of the dominant kernel from PMC counters, separate passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
#   PMC_BENCH_ARGS: extra bench.py arguments (e.g. "--config coco --batch 512 --precision bf16"); PMC_RAW: output file name
R=$GRAFT_REPO_ROOT
RAW=${PMC_RAW:-pmc_traffic_raw.json}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline $PMC_BENCH_ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py --no-graph --num-steps 3 --warmup 0 --no-cpu-baseline $PMC_BENCH_ARGS > /dev/null 2>&1
python3 - <<PY
import csv, glob, json, collections
out={}
for tag,d in (("FETCH_SIZE","/tmp/pmc_f"),("WRITE_SIZE","/tmp/pmc_w")):
    f=glob.glob(d+"/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!=tag: continue
        k=r["Kernel_Name"].split("(")[0]
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    out[tag]={k:{"launches":v[0],"sum":v[1]} for k,v in agg.items()}
json.dump(out,open("$R/gpurun_out/$RAW","w"),indent=1)
for tag in out:
    tot=sum(v["sum"] for v in out[tag].values())
    for k,v in sorted(out[tag].items(), key=lambda kv:-kv[1]["sum"])[:8]:
        print(tag, k[:70], v["launches"], "avg", v["sum"]/v["launches"], "share %.1f%%"%(100*v["sum"]/tot))
PY
