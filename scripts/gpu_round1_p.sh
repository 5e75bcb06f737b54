timeout 900 python -m pytest tests/test_gpu_roi_transformer.py tests/test_gpu_oriented_rcnn.py -m gpu -x -q 2>&1 | tail -15
