#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/r4_j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -5 $OUT/pytest_gpu.log
for wl in s2anet_train orcnn_train retinanet_infer roitrans_r50_train; do timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("'$wl'", round(d["value"],2), d["unit"], round(d["ms_per_step"],3), "ms")'; done
