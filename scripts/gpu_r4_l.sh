#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_s2anet.py tests/test_gpu_head_parity.py -q 2>&1 | tail -3
for p in 1024 4096 16384; do JDET_PACK_MAX_POS=$p timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("s2anet pack_max_pos='$p'", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms")'; done
JDET_PACK_MAX_POS=4096 timeout 600 python bench.py --workload retinanet_infer --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
