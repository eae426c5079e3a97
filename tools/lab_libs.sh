#!/bin/bash
# Lab (GPU box): stage-B time and the kernels matching a pattern for lab builds (tools/build_lab.sh tags).  usage: lab_libs.sh <grep pattern> <tag>...
set -e -o pipefail
PAT=$1; shift
OUT=gpurun_out/libs
mkdir -p $OUT
export TMPDIR=/tmp
for tag in "$@"; do
  lib=""
  [ "$tag" != default ] && lib=$PWD/tools/_lab/libibloc_$tag.so
  IBLOC_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$tag -o r -- python3 tools/perf_register.py 32 300 > $OUT/run_$tag.log 2>&1
  python3 tools/kstats.py $OUT/p_$tag 60 | grep -E "$PAT" > $OUT/k_$tag.txt || true
  grep "ms per step" $OUT/run_$tag.log >> $OUT/k_$tag.txt
  rm -rf $OUT/p_$tag
done
