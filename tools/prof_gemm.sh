#!/bin/bash
# dev helper (GPU box): rocprofv3 counters for one GEMM shape index / variant
# usage: tools/prof_gemm.sh <shape-index> <variant>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
IDX=${1:-0}; VAR=${2:-2}
rm -rf $R/gpurun_out/pmc_*
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_a -- $R/tools/bin/gemm_bench $IDX 5 $VAR > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_b -- $R/tools/bin/gemm_bench $IDX 5 $VAR > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $R/gpurun_out/pmc_c -- $R/tools/bin/gemm_bench $IDX 5 $VAR > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$R/gpurun_out/pmc_*/*/*counter_collection.csv")):
    rows=[r for r in csv.DictReader(open(f)) if 'gemm' in r['Kernel_Name']]
    agg=collections.defaultdict(list)
    for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print('%-32s n=%d mean=%.5g'%(k,len(v),sum(v)/len(v)))
PY
