python -m pytest tests/test_gpu_boxes.py tests/test_gpu_s2anet.py -m gpu -q -x 2>&1 | tail -30
