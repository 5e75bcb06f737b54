#!/bin/bash
# round 2, call O: backward with fused taps / single scan / striped 2x2 gather / kept workspace; wrw GEMM split-K variants
set -u
OUT=$PWD/gpurun_out/r2_o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py tests/test_gpu_dcn_arf.py tests/test_gpu_oriented_rcnn.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 200 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline > $OUT/bench_bwd.json 2> $OUT/bench_bwd.err
tail -1 $OUT/bench_bwd.json | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_bwd -o t -- python $OLDPWD/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline > $OUT/trace_bwd.log 2>&1
cd $OLDPWD
k=$(find $OUT/trace_bwd -name '*kernel_stats.csv' | head -1)
[ -n "$k" ] && head -9 $k | cut -c1-200 > $OUT/kernel_stats_bwd.csv
rm -rf $OUT/trace_bwd
cut -c1-170 $OUT/kernel_stats_bwd.csv
timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_s2anet.json 2> $OUT/bench_s2anet.err
tail -1 $OUT/bench_s2anet.json | cut -c1-220
