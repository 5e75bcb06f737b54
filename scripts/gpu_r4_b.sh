#!/bin/bash
# Round 4, second GPU session: plan + pool forward (parity, timing, counters), BN finish kernel, graph-replay diag.
set -u
R=$PWD
OUT=$R/gpurun_out/r4_b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_frozen_bn.py -x -q -k "sliced or golden or cfg0 or full_size or channels_last or riroi_vector or frozen or bn or bias" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
trace() {  # $1 = tag, rest = env
  tag=$1; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$tag -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 50 > $OUT/t_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/t_$tag/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    m=re.search(r"(roi_\w+_kernel)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("[$tag]", "; ".join("%s %.1f"%(k,sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5))
PY
}
trace default A=1
trace pred JDET_ROI_SLICED_PRED=1
trace b4 JDET_ROI_SLICED_BATCH=4
trace b4pred JDET_ROI_SLICED_BATCH=4 JDET_ROI_SLICED_PRED=1
trace b16 JDET_ROI_SLICED_BATCH=16
trace legacy JDET_ROI_FWD_LEGACY=1
pmc() {  # $1 = tag, $2 = counters, rest = env
  tag=$1; c=$2; shift; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p_$tag -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 10 --warmup 3 > $OUT/p_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections,re
for f in sorted(glob.glob("$OUT/p_$tag/**/*counter_collection.csv",recursive=True)):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r"(roi_\w+_kernel)", r["Kernel_Name"])
        if m: d[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print("[$tag]", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
}
pmc ea "TCC_EA0_RDREQ_sum TCC_READ_sum" A=1
pmc hit "TCC_HIT_sum TCC_MISS_sum" A=1
pmc fetch "FETCH_SIZE" A=1
pmc write "WRITE_SIZE" A=1
pmc tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" A=1
pmc sq "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" A=1
timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
# two ranks on one device, graph mode, probes at every hand-over (scripts/ddp_graph_diag.py)
timeout 900 python scripts/ddp_graph_diag.py orcnn 9 4 > $OUT/ddp_diag.log 2>&1
grep -E "^== run|RESULT|DISAGREE|GARBAGE|Error|error" $OUT/ddp_diag.log | cut -c1-400 | head -40
