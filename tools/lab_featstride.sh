#!/bin/bash
# Lab (GPU box): the matrix-core feature search with pass 1 on every s-th database chunk (IBL_FEAT_P1_STRIDE): kernel times of stage B of
# a bench-like step under rocprofv3 + the candidate counts (IBL_TIMING=2).
set -e -o pipefail
OUT=gpurun_out/fs
mkdir -p $OUT
export TMPDIR=/tmp
for s in "$@"; do
  IBL_FEAT_P1_STRIDE=$s timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$s -o r -- python3 tools/perf_register.py 32 300 > $OUT/run$s.log 2>&1
  python3 tools/kstats.py $OUT/p$s 40 | grep -E "feat|Name" > $OUT/k$s.txt || true
  rm -rf $OUT/p$s
  IBL_FEAT_P1_STRIDE=$s IBL_TIMING=2 timeout -k 10 200 python3 tools/perf_register.py 32 300 2>&1 | grep -E "candidates|ms per step" | tail -4 > $OUT/c$s.txt || true
done
