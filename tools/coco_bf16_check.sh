#!/bin/bash
# GPU box: BASELINE configs[4]'s per-GPU share (COCO-bits, B=512, T=20) in the opt-in bf16 mode: bench line, then kernel stats of the
# same command under rocprofv3 (eager launches: rocprofv3 7.2 cannot trace hipGraph replays) -> gpurun_out/coco/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/coco
mkdir -p $OUT
TAG=${1:-bx}
python3 $R/bench.py --config coco --batch 512 --num-steps 20 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/coco_B512_T20_bf16_$TAG.json 2> $OUT/coco_B512_T20_bf16_$TAG.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python3 $R/bench.py --config coco --batch 512 --num-steps 20 --precision bf16 --no-cpu-baseline --no-graph --warmup 0 --steps 1 > $OUT/coco_bf16_under_rocprof_$TAG.json 2> /dev/null
cp $(ls /tmp/kt5/*/*kernel_stats.csv | head -1) $OUT/coco_B512_T20_bf16_kernel_stats_$TAG.csv
echo "coco bf16 $TAG done"
