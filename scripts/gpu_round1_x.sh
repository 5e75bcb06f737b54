timeout 900 python -m pytest tests/test_gpu_oriented_rcnn.py tests/test_gpu_roi_transformer.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --workload orcnn_train --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
