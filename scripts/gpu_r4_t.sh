#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_s2anet.py tests/test_gpu_ddp_detectors.py tests/test_gpu_conv1x1.py tests/test_gpu_conv_igemm.py -x -q 2>&1 | tail -4
for v in 1 0 1 0; do
  echo "== JDET_CONV_BWD_STREAMS=$v"
  JDET_CONV_BWD_STREAMS=$v timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
for v in 1 0; do
  echo "== orcnn JDET_CONV_BWD_STREAMS=$v"
  JDET_CONV_BWD_STREAMS=$v timeout 600 python bench.py --workload orcnn_train --no-cpu-baseline --steps 20 --warmup 6 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
