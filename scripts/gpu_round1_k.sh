python -m pytest tests/test_gpu_s2anet.py -m gpu -q -x 2>&1 | tail -8
timeout 600 python bench.py --workload retinanet_infer --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-420
timeout 600 python bench.py --workload s2anet_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
