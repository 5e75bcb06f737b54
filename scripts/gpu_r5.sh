#!/bin/bash
# usage (GPU box): bash scripts/gpu_r5.sh <run> [args]   -- the round-5 measurement runs, one function per run
set -u
R=$PWD
export TMPDIR=/tmp

run_a() {   # first contact of the bottleneck family: parity, per-layer / per-block times, step A/B
  OUT=$R/gpurun_out/r5_a; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py layers blocks wgrad tiles 2>&1 | grep -v Warning | tee $OUT/timing.txt
  bash scripts/ab_step.sh -n 1 -s 20 "JDET_BOTTLENECK_FUSED=1" "JDET_BOTTLENECK_FUSED=0" 2>&1 | tee $OUT/ab.txt
}

run=${1:-}; [ $# -gt 0 ] && shift
case "$run" in
  a) run_$run "$@";;
  *) echo "usage: gpu_r5.sh {a} [args]"; exit 2;;
esac
