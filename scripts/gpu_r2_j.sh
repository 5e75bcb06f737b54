#!/bin/bash
set -u
OUT=gpurun_out/r2_j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_roi.log
tail -5 $OUT/pytest_roi.log
for path in roi_cl roi tile; do
  JDET_ROI_FWD_PATH=$path timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd $path: %.1f us frac %.3f' % (l['roofline']['kernel_ms']*1e3, l['roofline']['frac']))"
done
JDET_BENCH_NO_ORDER=1 JDET_ROI_FWD_PATH=roi_cl timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd roi_cl no-order: %.1f us frac %.3f' % (l['roofline']['kernel_ms']*1e3, l['roofline']['frac']))"
for lay in cl nchw; do
  JDET_BENCH_BWD_LAYOUT=$lay timeout 120 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bwd $lay: %.1f us frac %.3f' % (l['roofline']['kernel_ms']*1e3, l['roofline']['frac']))"
done
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1
echo "pytest-all rc=$?" >> $OUT/pytest_all.log
tail -5 $OUT/pytest_all.log
