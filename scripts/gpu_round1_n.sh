timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_oriented_rcnn.py -m gpu -x -q 2>&1 | tail -8
for m in 0 1; do
JDET_ROI_FWD_MODE=$m timeout 300 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-600
done
JDET_ROI_FWD_WAVES=8 timeout 300 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
