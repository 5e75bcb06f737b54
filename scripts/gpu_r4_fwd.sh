#!/bin/bash
# Round 4, forward: parity of the channel-sliced kernel, then kernel-trace averages and traffic counters of the
# product path under its tuning knobs and of the legacy (RoI-stationary) kernels.  -> gpurun_out/r4_fwd/
set -u
R=$PWD
OUT=$R/gpurun_out/r4_fwd; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_align.py -x -q -k "sliced or golden or cfg0 or full_size or channels_last or riroi_vector" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
trace() {  # $1 = tag, rest = env
  tag=$1; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$tag -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 50 > $OUT/t_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/t_$tag/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])): d[r["Kernel_Name"].split("::")[-1][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("[$tag]", "; ".join("%s %.1f"%(k,sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5), "|", open("$OUT/t_$tag.log").read().strip().split("\n")[-1][:160])
PY
}
trace default A=1
trace b4 JDET_ROI_SLICED_BATCH=4
trace b16 JDET_ROI_SLICED_BATCH=16
trace plain JDET_ROI_SLICED_STORE=1
trace sc1 JDET_ROI_SLICED_STORE=2
trace b16plain JDET_ROI_SLICED_BATCH=16 JDET_ROI_SLICED_STORE=1
trace legacy JDET_ROI_FWD_LEGACY=1
pmc() {  # $1 = tag, $2 = counters, rest = env
  tag=$1; c=$2; shift; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p_$tag -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 10 --warmup 3 > $OUT/p_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p_$tag/**/*counter_collection.csv",recursive=True)):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("::")[-1][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items():
        if "roi_" in k: print("[$tag]", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
}
for v in default legacy; do
  e="A=1"; [ $v = legacy ] && e="JDET_ROI_FWD_LEGACY=1"
  pmc ${v}_ea "TCC_EA0_RDREQ_sum TCC_READ_sum" $e
  pmc ${v}_hit "TCC_HIT_sum TCC_MISS_sum" $e
  pmc ${v}_fetch "FETCH_SIZE" $e
  pmc ${v}_write "WRITE_SIZE" $e
  pmc ${v}_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $e
  pmc ${v}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" $e
done
timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | tail -1
