#!/bin/bash
# Dev container: build libdsg variants with a timing experiment compiled into qkv_attn_bx_kernel (kernels_bx.hip DSG_QA_EXP / DSG_QA_OCC) -> tools/bin/ab/
# GPU box: tools/qa_exp.sh run -- per-variant time of the fused QKV + attention kernel at the three levels (tools/bx_bench.py BX_ONLY=qa)
cd "$(dirname "$0")/.."
VARS=${QA_EXPS:-"0 1 2 3 4 0o3 0o2"}
if [ "$1" = "run" ]; then
  for v in $VARS; do
    echo "=== DSG_QA_EXP=$v"
    BX_LIB=$PWD/tools/bin/ab/libdsg_qaexp$v.so BX_ITERS=10 BX_ONLY=qa python tools/bx_bench.py 2>&1 | grep fused
  done
  exit 0
fi
mkdir -p tools/bin/ab
cd diffusesg_amd/csrc
for v in $VARS; do
  e=${v%o*}; o=4; [[ $v == *o* ]] && o=${v#*o}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DDSG_QA_EXP=$e -DDSG_QA_OCC=$o -c kernels_bx.hip -o /tmp/kernels_bx_qaexp$v.o &
done
wait
for v in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/ab/libdsg_qaexp$v.so kernels.o kernels_lp.o /tmp/kernels_bx_qaexp$v.o train_kernels.o dsg_api.o
done
ls -la ../../tools/bin/ab/ | grep qaexp
