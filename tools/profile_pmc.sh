#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic of the GEMM kernel from two separate rocprofv3 --pmc passes of bench.py
# (FETCH_SIZE, WRITE_SIZE; no trace domains beside the counters), summarised by tools/pmc_summary.py.
set -e -o pipefail
OUT=gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    tag=$(echo $c | tr A-Z a-z | cut -d_ -f1)
    timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -o $tag -- python3 bench.py --cpu-frames 0 --steps 2 --warmup 1 > $OUT/bench_$tag.json
    f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
    head -1 $f > $OUT/${tag}_counter_collection_ibl_kernels.csv
    grep 'ibl_' $f >> $OUT/${tag}_counter_collection_ibl_kernels.csv
    rm -rf $OUT/$tag
done
python3 tools/pmc_summary.py $OUT/fetch_counter_collection_ibl_kernels.csv $OUT/write_counter_collection_ibl_kernels.csv $OUT/gemm_pmc.json
# MFMA utilisation and wave-time split (one more pass: 6 SQ counters + GRBM_GUI_ACTIVE)
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/mfma -o mfma -- python3 bench.py --cpu-frames 0 --steps 2 --warmup 1 > $OUT/bench_mfma.json
f=$(find $OUT/mfma -name '*counter_collection.csv' | head -1)
head -1 $f > $OUT/mfma_counter_collection_ibl_kernels.csv
grep 'ibl_' $f >> $OUT/mfma_counter_collection_ibl_kernels.csv
rm -rf $OUT/mfma
python3 tools/pmc_mfma_summary.py $OUT/mfma_counter_collection_ibl_kernels.csv $OUT/gemm_mfma_pmc.json
