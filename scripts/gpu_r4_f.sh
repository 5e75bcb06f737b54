#!/bin/bash
# Round 4: backward producer rewrite (independent atomics, scan folded in), RiRoIAlign static mix, full-size pins against
# the reference's kernels, reference-kernel timings, S2ANet step after the BN finish change, two-rank graph test x3
set -u
R=$PWD
OUT=$R/gpurun_out/r4_f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py tests/test_gpu_reference_kernels.py -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
trace() {  # $1 = tag, $2 = workload, rest = env
  tag=$1; wl=$2; shift; shift
  (cd /tmp && env JDET_BENCH_CHECKSUM=1 "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$tag -o t -- python $R/bench.py --workload $wl --no-cpu-baseline --steps 50 > $OUT/t_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/t_$tag/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    m=re.search(r"((roi|csr|bwd|riroi)_\w+_kernel)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
cs=[l.strip() for l in open("$OUT/t_$tag.log") if l.startswith("checksum")]
tot=sum(sum(v[5:])/len(v[5:]) for k,v in d.items() if len(v)>5)
print("[$tag] total %.1f :"%tot, "; ".join("%s %.1f"%(k,sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5), "|", cs[-1] if cs else "")
PY
}
trace bwd roi_align_rotated_bwd A=1
trace riroi riroi_align A=1
for wl in roi_align_rotated_bwd riroi_align roi_align_rotated; do timeout 120 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["workload"], round(d["ms_per_step"]*1000,1),"us/step frac", round(d["roofline"]["frac"],3))'; done
# the reference's own RoIAlign kernels on this GPU (kernel durations from the trace)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_refk -o t -- python $R/scripts/refk_diag.py --time > $OUT/refk_time.log 2>&1)
tail -3 $OUT/refk_time.log
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/t_refk/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if len(v)>=10: print("  %-90s n=%d avg %.1f us"%(k,len(v),sum(v[2:])/len(v[2:])))
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("s2anet", d["value"], "img/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], d["roofline"]["kernel_ms"])'
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_ddp_detectors.py -q -k "graph" --runxfail > $OUT/ddp_$i.log 2>&1; tail -1 $OUT/ddp_$i.log; grep -E "^FAILED|AssertionError|replicas diverged" $OUT/ddp_$i.log | cut -c1-600 | head -6; done
