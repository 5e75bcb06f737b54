#!/bin/bash
set -u
R=$PWD
OUT=$R/gpurun_out/r4_h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/conv1x1_probe.py 2>&1 | grep -v Warning | tail -20
timeout 900 python -m pytest tests/test_gpu_ddp_detectors.py -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_roi_align.py -x -q 2>&1 | tail -2
