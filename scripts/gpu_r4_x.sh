#!/bin/bash
# round 4: fused SGD step with the clip folded in (JDET_FUSED_SGD): tests + S2ANet / Oriented R-CNN step A/B
set -u
timeout 900 python -m pytest tests/test_gpu_s2anet.py tests/test_gpu_ddp_detectors.py -q 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "== JDET_FUSED_SGD=$v"
  JDET_FUSED_SGD=$v timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
for v in 1 0; do
  echo "== orcnn JDET_FUSED_SGD=$v"
  JDET_FUSED_SGD=$v timeout 600 python bench.py --workload orcnn_train --no-cpu-baseline --steps 20 --warmup 6 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
