#!/bin/bash
# Lab (GPU box): stage-B time and ICP kernel sums for lab builds of the ICP schedule (tools/build_lab.sh tags as arguments)
set -e -o pipefail
OUT=gpurun_out/icp
mkdir -p $OUT
export TMPDIR=/tmp
for tag in "$@"; do
  lib=""
  [ "$tag" != default ] && lib=$PWD/tools/_lab/libibloc_$tag.so
  IBLOC_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$tag -o r -- python3 tools/perf_register.py 32 300 > $OUT/run_$tag.log 2>&1
  python3 tools/kstats.py $OUT/p_$tag 60 | grep -E "icp" > $OUT/k_$tag.txt || true
  grep "ms per step" $OUT/run_$tag.log >> $OUT/k_$tag.txt
  rm -rf $OUT/p_$tag
done
