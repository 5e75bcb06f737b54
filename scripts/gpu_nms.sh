#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r3_nms; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_iou_nms.py tests/test_gpu_poly.py -x -q -m gpu 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python bench.py --workload nms_rotated --no-cpu-baseline --no-secondary --steps 50 > $OUT/bench.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/t/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])): d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    if len(v)>5: print("%-60s n=%d avg %.1f us"%(k,len(v),sum(v[5:])/len(v[5:])))
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log
