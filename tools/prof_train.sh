#!/bin/bash
# GPU box: rocprofv3 kernel stats of a few training iterations at the VG shape -> gpurun_out/train_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- python3 $R/tools/time_train.py > $R/gpurun_out/time_train_under_rocprof.txt 2>&1
cp $(ls /tmp/ktt/*/*kernel_stats.csv | head -1) $R/gpurun_out/train_kernel_stats.csv
