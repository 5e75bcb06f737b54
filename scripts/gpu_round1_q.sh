timeout 600 python -m pytest tests/test_gpu_s2anet.py tests/test_gpu_roi_transformer.py -m gpu -x -q 2>&1 | tail -4
for w in orcnn_train roitrans_r50_train; do
timeout 900 python bench.py --workload $w --batch 2 --steps 8 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
timeout 900 python bench.py --workload roitrans_train --batch 4 --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
