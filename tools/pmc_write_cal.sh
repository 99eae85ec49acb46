#!/bin/bash
# GPU box: WRITE_SIZE counter vs bytes actually written, per store shape (tools/write_size_cal.cpp) -> stdout
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_cal
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_cal -- $R/tools/bin/write_size_cal
python3 - <<PY
import csv, glob
f=glob.glob("/tmp/pmc_cal/*/*counter_collection.csv")[0]
want={"k_dword_row":256<<20,"k_dword_gemm":(( (256<<20)//(128*96*4))*128*96*4),"k_dwordx4":256<<20}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]!="WRITE_SIZE": continue
    k=r["Kernel_Name"].split("(")[0]
    if k in want:
        v=float(r["Counter_Value"])
        print("%-14s WRITE_SIZE %12.1f KiB  written %10.1f KiB  ratio %.4f" % (k, v, want[k]/1024, v*1024/want[k]))
PY
