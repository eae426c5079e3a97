#!/bin/bash
# Runs on the GPU box (via gpurun): the bench lines and rocprofv3 summaries that are copied into profiles/rNN/.
set -e -o pipefail
OUT=gpurun_out/round
mkdir -p $OUT
export TMPDIR=/tmp
# the driver's command (T: M = 10 000, DINOv2-B) with the CPU baseline leg, then the same steps back to back
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json
timeout -k 10 400 python3 bench.py --sequential --cpu-frames 0 > $OUT/bench_sequential.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --cpu-frames 0 > $OUT/bench_default_under_rocprof.json
python3 tools/step_breakdown.py $OUT/prof 32 > $OUT/bench_default_last_step_breakdown.txt
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
rm -rf $OUT/prof
# the other BASELINE configs that fit one GPU
timeout -k 10 400 python3 bench.py --config C2 --cpu-frames 0 > $OUT/bench_C2.json
timeout -k 10 500 python3 bench.py --config C3 > $OUT/bench_C3_dator.json
timeout -k 10 400 python3 bench.py --config C4 --cpu-frames 0 > $OUT/bench_C4_m50000.json
timeout -k 10 400 python3 bench.py --config C4 --cpu-frames 0 --shard-memory --comm rccl > $OUT/bench_C4_m50000_shard_rccl_world1.json
