#!/bin/bash
# usage: gpu_env.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- bench roi_align_rotated under each env set, print kernel averages
set -u
OUT=$PWD/gpurun_out/r3_env; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/e$i -o t -- python bench.py --workload roi_align_rotated --no-cpu-baseline --no-secondary --steps 50 > $OUT/e$i.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/e$i/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])): d[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("[$e]", "; ".join("%s %.1f"%(k.split("::")[-1][:22],sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5))
PY
done
