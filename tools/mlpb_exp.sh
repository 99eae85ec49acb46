#!/bin/bash
# Dev container: build libdsg variants with a timing experiment compiled into mlp_bx_kernel<96|192> (kernels_bx.hip DSG_MLPB_EXP) -> tools/bin/ab/
# GPU box: tools/mlpb_exp.sh run -- per-variant time of the level-0 / level-1 fused (proj +) MLP kernels (tools/bx_bench.py BX_ONLY=mlpb)
cd "$(dirname "$0")/.."
VARS=${MLPB_EXPS:-"0 1 2 3 4 5"}
if [ "$1" = "run" ]; then
  for v in $VARS; do
    echo "=== DSG_MLPB_EXP=$v"
    BX_LIB=$PWD/tools/bin/ab/libdsg_mlpbexp$v.so BX_ITERS=10 BX_ONLY=mlpb python tools/bx_bench.py 2>&1 | grep -E "L[01] "
  done
  exit 0
fi
mkdir -p tools/bin/ab
cd diffusesg_amd/csrc
for v in $VARS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DDSG_MLPB_EXP=$v -c kernels_bx.hip -o /tmp/kernels_bx_mlpbexp$v.o &
done
wait
for v in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/ab/libdsg_mlpbexp$v.so kernels.o kernels_lp.o /tmp/kernels_bx_mlpbexp$v.o train_kernels.o dsg_api.o
done
ls ../../tools/bin/ab/ | grep mlpbexp
