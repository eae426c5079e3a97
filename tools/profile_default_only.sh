set -e -o pipefail
OUT=gpurun_out/round
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --cpu-frames 0 > $OUT/bench_default_under_rocprof.json
python3 tools/step_breakdown.py $OUT/prof 32 > $OUT/bench_default_last_step_breakdown.txt
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
rm -rf $OUT/prof
