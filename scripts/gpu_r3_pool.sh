#!/bin/bash
# round 3: experimental register-cached forward -- parity + kernel averages vs the product kernel
set -u
OUT=$PWD/gpurun_out/r3_pool; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_experimental_pool.py tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest_pool.log 2>&1; echo "rc=$?" >> $OUT/pytest_pool.log
tail -5 $OUT/pytest_pool.log
bash scripts/gpu_env.sh "JDET_ROI_FWD_PATH=roi_cl" "JDET_ROI_FWD_PATH=pool"
