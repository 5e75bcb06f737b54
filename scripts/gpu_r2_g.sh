#!/bin/bash
set -u
OUT=gpurun_out/r2_h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_roi.log
tail -15 $OUT/pytest_roi.log
for wgs in 4 3 2; do
  echo "== wgs $wgs"
  JDET_ROI_TILE_PARTS=$wgs timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: %.1f us frac %.3f' % (l['roofline']['kernel_ms']*1e3, l['roofline']['frac']))"
done
JDET_ROI_TILE_DEBUG=1 timeout 300 python scripts/tile_timeline.py 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline.txt | head -50
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --workload roi_align_rotated --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r2_h/trace/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print(row['Name'][:60], row['Calls'], row['AverageNs'], row['MinNs'], row['MaxNs'])
PY
