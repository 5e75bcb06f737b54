#!/bin/bash
set -u
OUT=gpurun_out/r2_c
mkdir -p $OUT
for cpg in 8 4; do
  echo "== cpg $cpg"
  JDET_ROI_TILE_DEBUG=1 JDET_ROI_TILE_CPG=$cpg timeout 300 python scripts/tile_timeline.py 2>&1 | grep -v "^\[jdet tile\]" | tee $OUT/timeline_c$cpg.txt
  JDET_ROI_TILE_DEBUG=1 JDET_ROI_TILE_CPG=$cpg timeout 300 python scripts/tile_timeline.py 2>&1 | grep "^\[jdet tile\]" | head -1
done
