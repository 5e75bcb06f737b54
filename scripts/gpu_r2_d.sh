#!/bin/bash
set -u
OUT=gpurun_out/r2_d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_roi.log
tail -3 $OUT/pytest_roi.log
for cpg in 8 4 2; do
  echo "== cpg $cpg"
  JDET_ROI_TILE_CPG=$cpg timeout 300 python scripts/tile_timeline.py 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_c$cpg.txt
  JDET_ROI_TILE_CPG=$cpg timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench cpg $cpg: %.1f us frac %.3f' % (l['roofline']['kernel_ms']*1e3, l['roofline']['frac']))"
done
