#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r4_p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv_wgrad.py -x -q 2>&1 | tail -3
timeout 600 python scripts/conv_wgrad_timing.py ${KS:-0} 2>&1 | tee $OUT/wgrad_timing.jsonl
