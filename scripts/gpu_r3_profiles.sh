#!/bin/bash
# round 3 evidence run: smoke + default bench line + rocprofv3 kernel trace of the same command (steady-state step
# breakdown, roofline kernel stats), traffic counters of the roofline kernel (separate --pmc passes), kernel stats of the
# other hand-written kernels.  Output: gpurun_out/r3_prof/ (copy what is to be judged into profiles/).
set -u
R=$PWD
OUT=$R/gpurun_out/r3_prof
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-700
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/trace_default.log 2>&1
cd $R
f=$(find $OUT/trace_default -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 5 > $OUT/steady_state_s2anet.txt 2>&1
head -14 $OUT/steady_state_s2anet.txt | cut -c1-160
k=$(find $OUT/trace_default -name '*kernel_stats.csv' | head -1)
head -1 $k > $OUT/roofline_kernel_stats.csv
grep "roi_align_fwd_merged_kernel\|roi_order_kernel" $k >> $OUT/roofline_kernel_stats.csv
cut -c1-260 $OUT/roofline_kernel_stats.csv
rm -rf $OUT/trace_default
# traffic of the roofline kernel (default path), one counter set per pass
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_READ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  cd /tmp
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_fwd_$n -o p -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $OUT/pmc_fwd_$n.log 2>&1
  cd $R
done
python - <<PY > $OUT/roi_align_fwd_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_fwd_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if "roi_" in k:
        for c, v in sorted(cs.items()):
            print("%-70s %-32s mean %.6g over %d dispatches" % (k, c, sum(v) / len(v), len(v)))
PY
cat $OUT/roi_align_fwd_counters.txt | cut -c40-200
rm -rf $OUT/pmc_fwd_*/
for wl in roi_align_rotated_bwd box_iou_rotated nms_rotated retinanet_infer; do
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o t -- python $R/bench.py --workload $wl --no-cpu-baseline --no-secondary > $OUT/trace_$wl.log 2>&1
  cd $R
  k=$(find $OUT/trace_$wl -name '*kernel_stats.csv' | head -1)
  [ -n "$k" ] && head -14 $k | cut -c1-200 > $OUT/kernel_stats_$wl.csv
  grep -o '"ms_per_step": [0-9.]*' $OUT/trace_$wl.log | head -1
  rm -rf $OUT/trace_$wl
  echo "== $wl"; cut -c1-150 $OUT/kernel_stats_$wl.csv | head -8
done
for wl in orcnn_train roitrans_r50_train roitrans_train; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-secondary > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  python -c "
import json,sys
l=json.loads(open('$OUT/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', l['ms_per_step'], 'ms/step', l['value'], l['unit'])" 2>/dev/null || echo "$wl failed"
done
ls $OUT
