#!/bin/bash
# Dev container: build libdsg variants with a timing experiment compiled into qkv_attn_wx_kernel (kernels_bx.hip DSG_WX_EXP) -> tools/bin/ab/
# GPU box: tools/wx_exp.sh run -- per-variant time and phase clocks of the wave-per-unit QKV + attention kernel (tools/bx_bench.py BX_ONLY=wx)
cd "$(dirname "$0")/.."
VARS=${WX_EXPS:-"0 1 2 3 4"}
if [ "$1" = "run" ]; then
  for v in $VARS; do
    echo "=== DSG_WX_EXP=$v"
    DSG_WX_CLK=1 BX_LIB=$PWD/tools/bin/ab/libdsg_wxexp$v.so BX_ITERS=10 BX_ONLY=wx python tools/bx_bench.py 2>&1 | grep -E "fused|phases"
  done
  exit 0
fi
mkdir -p tools/bin/ab
cd diffusesg_amd/csrc
for v in $VARS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DDSG_WX_EXP=$v -c kernels_bx.hip -o /tmp/kernels_bx_wxexp$v.o &
done
wait
for v in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/ab/libdsg_wxexp$v.so kernels.o kernels_lp.o /tmp/kernels_bx_wxexp$v.o train_kernels.o dsg_api.o
done
ls ../../tools/bin/ab/ | grep wxexp
