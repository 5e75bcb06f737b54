#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/r4_i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv1x1.py tests/test_gpu_s2anet.py -q 2>&1 | tail -3
for g in 0 1; do JDET_CONV1X1_GEMM=$g timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("s2anet conv1x1_gemm='$g'", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms")'; done
for g in 0 1; do JDET_CONV1X1_GEMM=$g timeout 600 python bench.py --workload orcnn_train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("orcnn conv1x1_gemm='$g'", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms")'; done
