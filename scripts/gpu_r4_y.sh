#!/bin/bash
# round 4: LevelPack mask fused into the tower convs (JDET_PACK_FUSED_MASK) + RotationInvariantPooling kernels (JDET_RIP_KERNEL)
set -u
timeout 900 python -m pytest tests/test_gpu_dcn_arf.py tests/test_gpu_s2anet.py tests/test_gpu_conv_igemm.py tests/test_gpu_head_parity.py -q 2>&1 | tail -3
for cfg in "1 1" "0 0" "1 0" "1 1" "0 0"; do
  set -- $cfg
  echo "== JDET_PACK_FUSED_MASK=$1 JDET_RIP_KERNEL=$2"
  JDET_PACK_FUSED_MASK=$1 JDET_RIP_KERNEL=$2 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
