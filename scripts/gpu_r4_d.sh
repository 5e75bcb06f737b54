#!/bin/bash
# Round 4, forward experiments: workgroup size of the pool kernel, slice-planar map layout (L2 channel spread test)
set -u
R=$PWD
OUT=$R/gpurun_out/r4_d; mkdir -p $OUT
export TMPDIR=/tmp
trace() {  # $1 = tag, rest = env
  tag=$1; shift
  (cd /tmp && env JDET_BENCH_CHECKSUM=1 "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$tag -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 50 > $OUT/t_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/t_$tag/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    m=re.search(r"(roi_\w+_kernel)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
cs=[l.strip() for l in open("$OUT/t_$tag.log") if l.startswith("checksum")]
print("[$tag]", "; ".join("%s %.1f"%(k,sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5), "|", cs[-1] if cs else "")
PY
}
trace b4pred JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1
trace b4pred_w8 JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1 JDET_ROI_SLICED_WAVES=8
trace b4pred_w16 JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1 JDET_ROI_SLICED_WAVES=16
trace b4pred_w1 JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1 JDET_ROI_SLICED_WAVES=1
trace w16 JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_WAVES=16
trace planar JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_PLANAR=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1
trace planar_w16 JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_PLANAR=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1 JDET_ROI_SLICED_WAVES=16
trace legacy A=1
pmc() {  # $1 = tag, $2 = counters, rest = env
  tag=$1; c=$2; shift; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p_$tag -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 10 --warmup 3 > $OUT/p_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections,re
for f in sorted(glob.glob("$OUT/p_$tag/**/*counter_collection.csv",recursive=True)):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r"(roi_pool\w+_kernel)", r["Kernel_Name"])
        if m: d[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print("[$tag]", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
}
P="JDET_ROI_FWD_SLICED=1 JDET_ROI_SLICED_PLANAR=1 JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1"
pmc planar_ea "TCC_EA0_RDREQ_sum TCC_READ_sum" $P
pmc planar_hit "TCC_HIT_sum TCC_MISS_sum" $P
pmc planar_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" $P
pmc planar_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $P
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*STALL[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | cut -c1-3000
