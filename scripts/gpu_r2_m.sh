#!/bin/bash
set -u
OUT=gpurun_out/r2_m
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_oriented_rcnn.py tests/test_gpu_boxes.py tests/test_gpu_dcn_arf.py tests/test_gpu_roi_transformer.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -40 $OUT/pytest.log
timeout 600 python bench.py --workload orcnn_train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('orcnn_train: %.2f ms/step  %.1f img/s' % (l['ms_per_step'], l['value']))"
