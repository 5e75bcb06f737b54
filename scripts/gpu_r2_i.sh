#!/bin/bash
OUT=gpurun_out/r2_i
mkdir -p $OUT
JDET_ROI_TILE_DEBUG=1 timeout 300 python scripts/tile_timeline.py 2>&1 | grep -v "amdgpu.ids\|jdet tile" | tee $OUT/timeline.txt | head -60
