#!/bin/bash
# GPU box: the round's profile evidence in one call -> gpurun_out/profiles/ (copy what is to be judged into profiles/rN/).
#   1. rocprofv3 --kernel-trace --stats of the default bench workload (eager launches: rocprofv3 7.2 cannot trace hipGraph replays)
#   2. HBM traffic of the MFMA kernels: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/pmc_traffic.sh)
#   3. matrix-pipe utilisation per kernel: one --pmc pass (tools/pmc_mfma.sh)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline --no-graph --warmup 0 --steps 1 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(ls /tmp/kt/*/*kernel_stats.csv | head -1) $OUT/bench_default_nograph_kernel_stats.csv
cp $(ls /tmp/kt/*/*domain_stats.csv | head -1) $OUT/bench_default_nograph_domain_stats.csv 2>/dev/null
echo "kernel stats done" 
bash $R/tools/pmc_traffic.sh > $OUT/pmc_traffic_summary.txt 2>&1
cp $R/gpurun_out/pmc_traffic_raw.json $OUT/ 2>/dev/null
echo "traffic done"
bash $R/tools/pmc_mfma.sh > $OUT/pmc_mfma_summary.txt 2>&1
cp $R/gpurun_out/pmc_mfma_util.json $OUT/ 2>/dev/null
echo "mfma done"
DSG_PROFILE_VERBOSE=1 python3 $R/bench.py --steps 1 --warmup 1 --num-steps 50 --no-cpu-baseline > /dev/null 2> $OUT/forward_per_launch_B64.txt
echo "per-launch done"
# 4. configs[4]'s per-GPU share in the opt-in bf16 mode (COCO-bits, B=512, T=20): kernel stats and HBM traffic of gemm_bf16_kernel
rm -rf /tmp/kt5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python3 $R/bench.py --config coco --batch 512 --num-steps 20 --precision bf16 --no-cpu-baseline --no-graph --warmup 0 --steps 1 > $OUT/coco_bf16_under_rocprof.json 2> /dev/null
cp $(ls /tmp/kt5/*/*kernel_stats.csv | head -1) $OUT/coco_B512_T20_bf16_kernel_stats.csv
PMC_BENCH_ARGS="--config coco --batch 512 --precision bf16" PMC_RAW=pmc_traffic_raw_coco_bf16.json bash $R/tools/pmc_traffic.sh > $OUT/pmc_traffic_coco_bf16_summary.txt 2>&1
cp $R/gpurun_out/pmc_traffic_raw_coco_bf16.json $OUT/ 2>/dev/null
echo "coco bf16 done"
# 5. matrix-pipe / LDS counters of the bf16 pipeline's kernels (tools/pmc_lds.sh), and configs[3]'s per-GPU share (VG B = 256) HBM traffic
bash $R/tools/pmc_lds.sh > $OUT/pmc_lds_mfma_coco_bf16_summary.txt 2>&1
cp $R/gpurun_out/pmc_lds_coco_bf16.json $OUT/pmc_lds_mfma_coco_bf16.json 2>/dev/null
echo "coco bf16 counters done"
PMC_BENCH_ARGS="--batch 256" PMC_RAW=pmc_traffic_raw_vg_B256.json bash $R/tools/pmc_traffic.sh > $OUT/pmc_traffic_vg_B256_summary.txt 2>&1
cp $R/gpurun_out/pmc_traffic_raw_vg_B256.json $OUT/ 2>/dev/null
echo "vg B=256 traffic done"
