timeout 900 python -m pytest tests/test_gpu_iou_nms.py tests/test_gpu_oriented_rcnn.py -m gpu -x -q 2>&1 | tail -6
timeout 300 python bench.py --workload nms_rotated --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
timeout 600 python bench.py --workload retinanet_infer --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
