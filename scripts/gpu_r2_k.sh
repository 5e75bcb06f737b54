#!/bin/bash
set -u
OUT=gpurun_out/r2_l
mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_all.log 2>&1
echo "pytest-all rc=$?" >> $OUT/pytest_all.log
tail -30 $OUT/pytest_all.log
