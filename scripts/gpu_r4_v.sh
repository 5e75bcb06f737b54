#!/bin/bash
# round 4: the line-dedup forward (csrc/roi_align_line.h, JDET_ROI_FWD_LINE=1): parity suites, then bench + kernel trace A/B
set -u
R=$PWD; OUT=$R/gpurun_out/r4_v; mkdir -p $OUT; export TMPDIR=/tmp
JDET_ROI_FWD_LINE=1 timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_reference_kernels.py -q 2>&1 | tail -4
for v in ${LINES:-0 1 0 1}; do
  echo "== JDET_ROI_FWD_LINE=$v"
  JDET_ROI_FWD_LINE=$v timeout 300 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
for v in ${TRACE:-0 1}; do
  (cd /tmp && JDET_ROI_FWD_LINE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 50 > $OUT/t_$v.log 2>&1)
  k=$(find $OUT/t_$v -name '*kernel_stats.csv' | head -1); echo "LINE=$v"; head -4 $k | cut -c1-160
  rm -rf $OUT/t_$v
done
if [ "${PMC:-0}" = "1" ]; then
for v in ${TRACE:-0 1}; do
  for c in "TCC_EA0_RDREQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-20)
    (cd /tmp && JDET_ROI_FWD_LINE=$v timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p_${v}_$n -o p -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 10 --warmup 3 > $OUT/p_${v}_$n.log 2>&1)
  done
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p_${v}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "roi_align_fwd" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("LINE=$v", {k: round(sum(x)/len(x)) for k, x in sorted(agg.items())})
PY
  rm -rf $OUT/p_${v}_*/
done
fi
