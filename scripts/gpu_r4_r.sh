#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r4_r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv_wgrad.py tests/test_gpu_dcn_arf.py tests/test_gpu_conv_igemm.py tests/test_gpu_s2anet.py -x -q 2>&1 | tail -8
python scripts/conv_wgrad_timing.py 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['shape'], 'lib', d['lib_us'], 'own', d['own_us_ks0'])"
for cfg in "1 32768" "0 0" "1 0" "1 8192" "1 32768" "0 0"; do
  set -- $cfg
  echo "== JDET_CONV_WGRAD=$1 JDET_DCN_FUSED_TRAIN_MIN_POS=$2"
  JDET_CONV_WGRAD=$1 JDET_DCN_FUSED_TRAIN_MIN_POS=$2 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
