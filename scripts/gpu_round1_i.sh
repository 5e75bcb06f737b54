python -m pytest tests/test_gpu_s2anet.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python bench.py --workload s2anet_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-900
timeout 600 python bench.py --workload s2anet_train --steps 10 --warmup 3 --no-cpu-baseline --amp bf16 2>&1 | tail -1 | cut -c1-300
