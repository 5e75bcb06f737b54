#!/bin/bash
# round 4: pack / unpack as autograd Functions (JDET_PACK_FUNCTIONS) + channels-last ARF bank (JDET_ARF_CL): step A/B
set -u
for cfg in "1 1" "0 0" "1 1" "0 0"; do
  set -- $cfg
  echo "== JDET_PACK_FUNCTIONS=$1 JDET_ARF_CL=$2"
  JDET_PACK_FUNCTIONS=$1 JDET_ARF_CL=$2 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
