timeout 900 python -m pytest tests/test_gpu_boxes.py tests/test_gpu_s2anet.py tests/test_gpu_oriented_rcnn.py tests/test_gpu_roi_transformer.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --workload s2anet_train --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-220
