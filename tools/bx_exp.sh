#!/bin/bash
# Dev container: build libdsg variants with a timing experiment compiled into the bf16 GEMM (kernels_bx.hip DSG_BX_EXP) -> tools/bin/ab/
# GPU box: tools/bx_exp.sh run  -- runs tools/bx_bench.py against each variant (BX_LIB -> lib.load(path))
cd "$(dirname "$0")/.."
if [ "$1" = "run" ]; then
  for v in ${BX_EXPS:-0 1 2 3 4}; do
    echo "=== DSG_BX_EXP=$v"
    DSG_BX_DBG=$([ $v = 4 ] && echo 1) BX_LIB=$PWD/tools/bin/ab/libdsg_bxexp$v.so BX_ITERS=10 BX_ONLY=gemm python tools/bx_bench.py 2>&1 | grep -v "^B=\|amdgpu.ids"
  done
  exit 0
fi
mkdir -p tools/bin/ab
cd diffusesg_amd/csrc
for v in ${BX_EXPS:-0 1 2 3 4}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DDSG_BX_EXP=$v -c kernels_bx.hip -o /tmp/kernels_bx_exp$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/ab/libdsg_bxexp$v.so kernels.o kernels_lp.o /tmp/kernels_bx_exp$v.o train_kernels.o dsg_api.o
done
ls -la ../../tools/bin/ab/
