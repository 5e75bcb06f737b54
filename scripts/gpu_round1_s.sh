timeout 900 python -m pytest tests/test_gpu_s2anet.py tests/test_gpu_dcn_arf.py tests/test_gpu_roi_align.py -m gpu -x -q 2>&1 | tail -6
for g in 1 0; do
JDET_TRAIN_GRAPH=$g timeout 600 python bench.py --workload s2anet_train --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
done
