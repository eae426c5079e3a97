#!/bin/bash
# Runs on the GPU box (via gpurun): part A of the round-3 profile set -- the driver's command, its rocprofv3 kernel summary, the
# back-to-back form and the adjacent-objects variant.  Copy gpurun_out/round3/* into profiles/r03/.
set -e -o pipefail
OUT=gpurun_out/round3
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --cpu-frames 0 --ransac-budget 0 > $OUT/bench_default_under_rocprof.json
python3 tools/step_breakdown.py $OUT/prof 40 > $OUT/bench_default_last_step_breakdown.txt
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
rm -rf $OUT/prof
timeout -k 10 300 python3 bench.py --sequential --cpu-frames 0 --ransac-budget 0 > $OUT/bench_sequential.json
