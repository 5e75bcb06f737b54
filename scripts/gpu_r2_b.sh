#!/bin/bash
# round 2, call B: tile kernel after the scan / chunk-rotation changes: parity, sweep, PMC passes
set -u
OUT=gpurun_out/r2_b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_roi.log
tail -3 $OUT/pytest_roi.log
for shape in 0 1 3; do
  for cpg in 2 4 8; do
    JDET_ROI_TILE_SHAPE=$shape JDET_ROI_TILE_CPG=$cpg timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline > $OUT/bench_tile_s${shape}_c${cpg}.json 2>$OUT/bench_tile_s${shape}_c${cpg}.err
    python - <<PY
import json
try:
    l=json.loads(open("$OUT/bench_tile_s${shape}_c${cpg}.json").read().strip().splitlines()[-1])
    print("shape $shape cpg $cpg: %.1f us  frac %.3f" % (l["roofline"]["kernel_ms"]*1e3, l["roofline"]["frac"]))
except Exception as e:
    print("shape $shape cpg $cpg: FAILED", e)
PY
  done
done
JDET_ROI_TILE_CPG=8 bash scripts/gpu_pmc.sh r2_b/pmc_tile_c8 --workload roi_align_rotated > $OUT/pmc_c8.log 2>&1
tail -40 $OUT/pmc_c8.log
