#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r4_o; mkdir -p $OUT; export TMPDIR=/tmp
for kb in 0 20 26 36 52; do
  (cd /tmp && JDET_ROI_BWD_GATHER_LDS_KB=$kb timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$kb -o t -- python $R/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --steps 50 > $OUT/t_$kb.log 2>&1)
  python - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/t_$kb/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    m=re.search(r"((csr|bwd)_\w+_kernel)", r["Kernel_Name"])
    if m: d[m.group(1)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("lds_kb=$kb", "; ".join("%s %.1f"%(k,sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5))
PY
done
for kb in 0 26; do
(cd /tmp && JDET_ROI_BWD_GATHER_LDS_KB=$kb timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_READ_sum --output-format csv -d $OUT/p_$kb -o t -- python $R/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --steps 10 --warmup 3 > $OUT/p_$kb.log 2>&1)
python - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/p_$kb/**/*counter_collection.csv",recursive=True):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "gather" in r["Kernel_Name"]: d["gather"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("lds_kb=$kb gather", {c: round(sum(x)/len(x)) for c,x in d["gather"].items()})
PY
done
