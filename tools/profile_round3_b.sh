#!/bin/bash
# Part B: the adjacent-objects variant of T (objects on a 0.7 m grid: neighbourhoods of different instances overlap, so cross-instance
# feature groups are recomputed per job) and the other BASELINE configs that fit one GPU.
set -e -o pipefail
OUT=gpurun_out/round3
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --spacing 0.7 --cpu-frames 0 --ransac-budget 0 > $OUT/bench_T_adjacent_spacing0.7.json
timeout -k 10 300 python3 bench.py --config C2 --cpu-frames 0 --ransac-budget 0 > $OUT/bench_C2.json
timeout -k 10 300 python3 bench.py --config C4 --cpu-frames 0 > $OUT/bench_C4_m50000.json
timeout -k 10 300 python3 bench.py --config C4 --cpu-frames 0 --layout sharded > $OUT/bench_C4_m50000_shard_rccl_world1.json
