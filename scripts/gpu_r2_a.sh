#!/bin/bash
# round 2, call A: parity of the tile-stationary RoIAlign forward + first timings of its shapes
set -u
OUT=gpurun_out/r2_a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_roi.log
tail -5 $OUT/pytest_roi.log
for shape in 0 1 2 3; do
  for cpg in 1 2 4 8; do
    JDET_ROI_TILE_SHAPE=$shape JDET_ROI_TILE_CPG=$cpg timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline > $OUT/bench_tile_s${shape}_c${cpg}.json 2>$OUT/bench_tile_s${shape}_c${cpg}.err
    python - <<PY
import json
try:
    l=json.loads(open("$OUT/bench_tile_s${shape}_c${cpg}.json").read().strip().splitlines()[-1])
    print("shape $shape cpg $cpg: %.1f us  frac %.3f" % (l["roofline"]["kernel_ms"]*1e3, l["roofline"]["frac"]))
except Exception as e:
    print("shape $shape cpg $cpg: FAILED", e)
PY
  done
done
JDET_ROI_FWD_PATH=tile_exact timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline > $OUT/bench_tile_exact.json 2>&1
JDET_ROI_FWD_PATH=roi timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline > $OUT/bench_roi.json 2>&1
tail -c 400 $OUT/bench_tile_exact.json; echo; tail -c 400 $OUT/bench_roi.json; echo
JDET_BENCH_BWD_LAYOUT=cl timeout 120 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline > $OUT/bench_bwd_cl.json 2>&1
JDET_BENCH_BWD_LAYOUT=nchw timeout 120 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline > $OUT/bench_bwd_nchw.json 2>&1
tail -c 300 $OUT/bench_bwd_cl.json; echo; tail -c 300 $OUT/bench_bwd_nchw.json; echo
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_tile -o tile -- python $GRAFT_REPO_ROOT/bench.py --workload roi_align_rotated --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_tile.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof_tile -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_tile -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
