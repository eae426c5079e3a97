#!/bin/bash
# Runs on the GPU box (via gpurun): the round-4 profile set.  usage: tools/profile_round4.sh a|b|b1|b2|c   (copy gpurun_out/round4/* into profiles/r04/)
#   a  the driver's command (bench.py: value, value_min/max, value_with_h2d, value_adjacent, roofline, cpu_baseline), its rocprofv3 kernel
#      summary, the last-step breakdown, the back-to-back form
#   b  the other BASELINE configs that fit one GPU: C1, C2, C3 (DATOR), C4 slice (also through the sharded exchange at world 1)
#   c  PMC passes (counters in their own runs: no trace domains beside them) of the encoder forward and of stage B; GEMM per-tile stamps
set -e -o pipefail
OUT=gpurun_out/round4
mkdir -p $OUT
export TMPDIR=/tmp
case "$1" in
a)
    timeout -k 10 700 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --cpu-frames 0 --ransac-budget 0 --adjacent-spacing 0 --no-h2d --repeats 1 > $OUT/bench_default_under_rocprof.json
    python3 tools/step_breakdown.py $OUT/prof 40 > $OUT/bench_default_last_step_breakdown.txt
    cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
    rm -rf $OUT/prof
    timeout -k 10 300 python3 bench.py --sequential --cpu-frames 0 --ransac-budget 0 --adjacent-spacing 0 --no-h2d --repeats 1 > $OUT/bench_sequential.json
    ;;
b|b1)
    timeout -k 10 300 python3 bench.py --config C1 > $OUT/bench_C1.json
    timeout -k 10 300 python3 bench.py --config C2 --cpu-frames 0 --ransac-budget 0 > $OUT/bench_C2.json
    timeout -k 10 500 python3 bench.py --config C3 --ransac-budget 0 > $OUT/bench_C3_dator.json
    ;;&
b|b2)
    timeout -k 10 300 python3 bench.py --config C4 --cpu-frames 0 > $OUT/bench_C4_m50000.json
    timeout -k 10 700 python3 bench.py --config C4 --cpu-frames 0 --layout sharded > $OUT/bench_C4_m50000_shard_rccl_world1.json
    ;;
c)
    mkdir -p $OUT/pmc
    keep() { f=$(find $1 -name '*counter_collection.csv' | head -1); head -1 $f > $2; grep 'ibl_' $f >> $2; rm -rf $1; }
    for c in FETCH_SIZE WRITE_SIZE; do
        tag=$(echo $c | tr A-Z a-z | cut -d_ -f1)
        timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/t -o p -- python3 tools/perf_vit.py dinov2_vitb14 224 > /dev/null
        keep $OUT/pmc/t $OUT/pmc/vit_${tag}_counter_collection_ibl_kernels.csv
        timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/t -o p -- python3 tools/perf_register.py > /dev/null
        keep $OUT/pmc/t $OUT/pmc/reg_${tag}_counter_collection_ibl_kernels.csv
    done
    timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc/t -o p -- python3 tools/perf_vit.py dinov2_vitb14 224 > /dev/null
    keep $OUT/pmc/t $OUT/pmc/vit_mfma_counter_collection_ibl_kernels.csv
    timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc/t -o p -- python3 tools/perf_register.py > /dev/null
    keep $OUT/pmc/t $OUT/pmc/reg_wave_counter_collection_ibl_kernels.csv
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc/t -o p -- python3 tools/perf_register.py > $OUT/pmc/perf_register.txt
    cp $(find $OUT/pmc/t -name '*kernel_stats.csv' | head -1) $OUT/pmc/reg_kernel_stats.csv
    rm -rf $OUT/pmc/t
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc/t -o p -- python3 tools/perf_vit.py dinov2_vitb14 224 > $OUT/pmc/perf_vit.txt
    cp $(find $OUT/pmc/t -name '*kernel_stats.csv' | head -1) $OUT/pmc/vit_kernel_stats.csv
    rm -rf $OUT/pmc/t
    python3 tools/pmc_round3_summary.py $OUT/pmc $OUT
    python3 tools/perf_gemm.py --stamps > $OUT/gemm_tile_stamps.txt 2>&1 || true
    ;;
a|b1) ;;
*) echo "usage: $0 a|b|b1|b2|c"; exit 2 ;;
esac
