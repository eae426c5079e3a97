#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic and MFMA utilisation of the GEMM kernel from separate rocprofv3 --pmc passes (no
# trace domains beside the counters).  The profiled command is the encoder forward of one 224-crop batch (tools/perf_vit.py
# dinov2_vitb14 224: the 50 GEMM launches of a bench step's embed stage, nothing else) -- the whole bench under counter collection
# (every dispatch serialised, a 10 000-instance memory to set up) does not finish inside a gpurun call.
set -e -o pipefail
OUT=gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    tag=$(echo $c | tr A-Z a-z | cut -d_ -f1)
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -o $tag -- python3 tools/perf_vit.py dinov2_vitb14 224 > $OUT/vit_$tag.txt
    f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
    head -1 $f > $OUT/${tag}_counter_collection_ibl_kernels.csv
    grep 'ibl_' $f >> $OUT/${tag}_counter_collection_ibl_kernels.csv
    rm -rf $OUT/$tag
done
python3 tools/pmc_summary.py $OUT/fetch_counter_collection_ibl_kernels.csv $OUT/write_counter_collection_ibl_kernels.csv $OUT/gemm_pmc.json
# MFMA utilisation and wave-time split (one more pass: 6 SQ counters + GRBM_GUI_ACTIVE)
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/mfma -o mfma -- python3 tools/perf_vit.py dinov2_vitb14 224 > $OUT/vit_mfma.txt
f=$(find $OUT/mfma -name '*counter_collection.csv' | head -1)
head -1 $f > $OUT/mfma_counter_collection_ibl_kernels.csv
grep 'ibl_' $f >> $OUT/mfma_counter_collection_ibl_kernels.csv
rm -rf $OUT/mfma
python3 tools/pmc_mfma_summary.py $OUT/mfma_counter_collection_ibl_kernels.csv $OUT/gemm_mfma_pmc.json
