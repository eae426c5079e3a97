#!/bin/bash
# Part C: config C3 (DATOR) and the PMC passes.  Counters are collected in their own runs (no trace domains beside --kernel-trace):
#   * encoder forward of one 224-crop batch (tools/perf_vit.py): FETCH_SIZE, WRITE_SIZE, MFMA busy + wave-time split of the GEMM
#   * stage B of a bench step on a 300-instance memory (tools/perf_register.py): FETCH_SIZE, WRITE_SIZE, wave-time split and VALU
#     utilisation of the registration kernels; an un-profiled kernel trace of the same command gives the durations
set -e -o pipefail
OUT=gpurun_out/round3
mkdir -p $OUT/pmc
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --config C3 --ransac-budget 0 > $OUT/bench_C3_dator.json
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
