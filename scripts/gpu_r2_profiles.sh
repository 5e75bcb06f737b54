#!/bin/bash
# round 2 evidence run: smoke + default bench line + rocprofv3 stats of the same command + steady-state step breakdown,
# PMC passes of the roofline kernel, kernel stats of the other hand-written kernels, MFMA counters of the train step
set -u
R=$PWD
OUT=$R/gpurun_out/r2_prof
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-600
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_default -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/trace_default.log 2>&1
cd $R
f=$(find $OUT/trace_default -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 5 > $OUT/steady_state_s2anet.txt 2>&1
head -12 $OUT/steady_state_s2anet.txt | cut -c1-160
k=$(find $OUT/trace_default -name '*kernel_stats.csv' | head -1)
head -1 $k > $OUT/roofline_kernel_stats.csv
grep "roi_align_fwd_merged_kernel\|roi_order_kernel" $k >> $OUT/roofline_kernel_stats.csv
cut -c1-260 $OUT/roofline_kernel_stats.csv
rm -rf $OUT/trace_default
# PMC passes of the roofline kernel (default path)
bash scripts/gpu_pmc.sh r2_prof/pmc_fwd --workload roi_align_rotated > $OUT/pmc_fwd.log 2>&1
grep "FETCH_SIZE\|WRITE_SIZE\|TCC_\|LDS_\|TCP_" $OUT/pmc_fwd/summary.txt | cut -c60-200
rm -rf $OUT/pmc_fwd/trace $OUT/pmc_fwd/pmc_*
# kernel stats of the other hand-written kernels
for wl in roi_align_rotated_bwd box_iou_rotated nms_rotated; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$wl -o t -- python $R/bench.py --workload $wl --no-cpu-baseline --no-secondary > $OUT/trace_$wl.log 2>&1
  cd $R
  k=$(find $OUT/trace_$wl -name '*kernel_stats.csv' | head -1)
  [ -n "$k" ] && head -12 $k | cut -c1-200 > $OUT/kernel_stats_$wl.csv
  rm -rf $OUT/trace_$wl
  echo "== $wl"; cut -c1-150 $OUT/kernel_stats_$wl.csv | head -8
done
# backward traffic
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  timeout 300 rocprofv3 --pmc $c -f csv -d $OUT/pmc_bwd_$c -o p -- python $R/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $OUT/pmc_bwd_$c.log 2>&1
  cd $R
done
python scripts/summarize_prof.py $OUT > $OUT/pmc_bwd_summary.txt 2>&1 || true
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_bwd_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:50]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print("bwd %s %-50s mean %.1f over %d" % (c, k, sum(v) / len(v), len(v)))
PY
rm -rf $OUT/pmc_bwd_FETCH_SIZE $OUT/pmc_bwd_WRITE_SIZE
# MFMA counters over the S2ANet step (own pass, kernel-trace only)
cd /tmp
for c in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --pmc $c -f csv -d $OUT/mfma_$n -o p -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/mfma_$n.log 2>&1 || echo "pmc $c failed"
done
cd $R
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/mfma_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:90]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k] += 1
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:25]
with open("$OUT/mfma_counters_s2anet.txt", "w") as o:
    o.write("kernel | dispatch rows | " + " | ".join(sorted({c for _, v in rows for c in v})) + "\n")
    for k, v in rows:
        o.write("%s | %d | %s\n" % (k, cnt[k], " | ".join("%s=%.4g" % (c, v[c]) for c in sorted(v))))
print(open("$OUT/mfma_counters_s2anet.txt").read()[:3000])
PY
rm -rf $OUT/mfma_SQ*
ls $OUT
