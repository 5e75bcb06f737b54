#!/bin/bash
# usage: gpu_counters.sh "<pmc counters>" "ENV=.." ["ENV=.." ...] -- per-kernel counter averages of the roi_align_rotated bench
set -u
OUT=$PWD/gpurun_out/r3_pmc; mkdir -p $OUT
export TMPDIR=/tmp
pmc=$1; shift
i=0
cd /tmp
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o t -- python $OLDPWD/bench.py --workload roi_align_rotated --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $OUT/p$i.log 2>&1
  python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p$i/**/*counter_collection.csv",recursive=True)):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("::")[-1][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items():
        if "roi_" in k: print("[$e]", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
