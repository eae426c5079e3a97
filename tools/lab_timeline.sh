#!/bin/bash
# Lab (GPU box): launch-by-launch timeline of one stage-B batch (tools/perf_register.py under rocprofv3 --kernel-trace)
set -e -o pipefail
OUT=gpurun_out/tl
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o r -- python3 tools/perf_register.py 32 300 > $OUT/run.log 2>&1
python3 tools/step_timeline.py $OUT/p > $OUT/timeline.txt
rm -rf $OUT/p
