set -x
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --workload retinanet_infer --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
timeout 900 python bench.py --workload orcnn_train --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
timeout 900 python bench.py --workload s2anet_train --amp bf16 --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
