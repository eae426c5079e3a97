#!/bin/bash
# Lab (GPU box): feature time of one bench step's detections (tools/perf_features.py) over the tile-grid knobs IBL_KNN_SAFETY x IBL_KNN_RHO
set -e -o pipefail
OUT=gpurun_out/knn
mkdir -p $OUT
for rho in 2 3 4; do
  for sf in 0.6 0.8 1.0 1.25 1.6; do
    echo "rho $rho safety $sf: $(IBL_KNN_RHO=$rho IBL_KNN_SAFETY=$sf IBL_KNN_DEBUG=1 timeout -k 10 120 python3 tools/perf_features.py 2>&1 | tail -2 | tr '\n' ' ')" >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
