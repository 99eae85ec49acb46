#!/bin/bash
# Dev container: build libdsg variants with a timing experiment compiled into mlp384d_bx_kernel (kernels_bx.hip DSG_M384_EXP) -> tools/bin/ab/
# GPU box: tools/m384_exp.sh run -- per-variant time and in-kernel phase clocks of the C = 384 fused (proj +) MLP (tools/bx_bench.py)
cd "$(dirname "$0")/.."
if [ "$1" = "run" ]; then
  for v in ${M384_EXPS:-0 1 2 3 5}; do
    echo "=== DSG_M384_EXP=$v"
    DSG_M384_SKEW=${M384_SKEW:--1} DSG_M384_CLK=1 BX_LIB=$PWD/tools/bin/ab/libdsg_m384exp$v.so BX_ITERS=10 BX_ONLY=${M384_ONLY:-mlp384} python tools/bx_bench.py 2>&1 | grep -v "^B=\|amdgpu.ids"
  done
  exit 0
fi
mkdir -p tools/bin/ab
cd diffusesg_amd/csrc
for v in ${M384_EXPS:-0 1 2 3 5}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DDSG_M384_EXP=${v%p*} $( [[ $v == *p0 ]] && echo -DDSG_M384_PRIO=0 ) -c kernels_bx.hip -o /tmp/kernels_bx_m384exp$v.o &
done
wait
for v in ${M384_EXPS:-0 1 2 3 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/ab/libdsg_m384exp$v.so kernels.o kernels_lp.o /tmp/kernels_bx_m384exp$v.o train_kernels.o dsg_api.o
done
ls -la ../../tools/bin/ab/ | grep m384
