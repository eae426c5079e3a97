#!/bin/bash
# Runs on the GPU box (via gpurun): the bench lines and rocprofv3 summaries that are copied into profiles/rNN/.
set -e -o pipefail
OUT=gpurun_out/round
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json
timeout -k 10 300 python3 bench.py --sequential --cpu-frames 0 > $OUT/bench_sequential.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --cpu-frames 0 > $OUT/bench_default_under_rocprof.json
python3 tools/step_breakdown.py $OUT/prof 32 > $OUT/bench_default_last_step_breakdown.txt
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
rm -rf $OUT/prof
timeout -k 10 500 python3 bench.py --memory 10000 --cpu-frames 0 > $OUT/bench_T_memory10000.json
