#!/bin/bash
# Round 4: backward with 4x4 patches (parity + A/B against the 2x2 path), RiRoIAlign forward after the mix rewrite
set -u
R=$PWD
OUT=$R/gpurun_out/r4_e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python -m pytest tests/test_gpu_reference_kernels.py -x -q -k "roi" > $OUT/pytest_ref.log 2>&1
echo "pytest ref rc=$?"; tail -3 $OUT/pytest_ref.log
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
trace bwd4 roi_align_rotated_bwd A=1
trace bwd2 roi_align_rotated_bwd JDET_ROI_BWD_PATCH=2
trace riroi riroi_align A=1
trace fwd roi_align_rotated A=1
for wl in roi_align_rotated_bwd riroi_align roi_align_rotated; do timeout 120 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["workload"], round(d["ms_per_step"]*1000,1),"us/step frac", round(d["roofline"]["frac"],3))'; done
