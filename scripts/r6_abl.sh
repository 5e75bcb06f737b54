#!/bin/bash
# ablations of the staged forward at the north-star point (JDET_ROI_STAGE_ABL bits: 1 prologue only, 2 no DMA, 4 no compute, 8 no stores)
for cpp in 64 32; do
for abl in 0 1 2 4 6 8; do
  echo "== CPP=$cpp ABL=$abl"
  JDET_ROI_STAGE_CPP=$cpp JDET_ROI_STAGE_ABL=$abl timeout 200 python scripts/r6_stage.py time 100 2>&1 | grep -v amdgpu.ids | grep "staged\|north"
done
done
