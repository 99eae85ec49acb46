#!/bin/bash
# GPU box: kernel stats of training iterations at the VG shape, B=64 (tools/time_train.py restricted to B=64) -> gpurun_out/train/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/train
mkdir -p $OUT
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktt
TT_BATCHES=64 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- python3 $R/tools/time_train.py > $OUT/time_train_$TAG.txt 2>&1
cp $(ls /tmp/ktt/*/*kernel_stats.csv | head -1) $OUT/train_iteration_vg_B64_kernel_stats_$TAG.csv
TT_BATCHES=64 python3 $R/tools/time_train.py >> $OUT/time_train_$TAG.txt 2>&1
tail -3 $OUT/time_train_$TAG.txt
