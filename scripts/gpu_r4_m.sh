#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r4_m; mkdir -p $OUT; export TMPDIR=/tmp
for e in 0 1 0 1; do JDET_BENCH_CHECKSUM=1 JDET_ROI_FWD_EXACT=$e timeout 120 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>$OUT/err_$e.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("exact='$e'", round(d["ms_per_step"]*1000,2),"us/step", round(d["roofline"]["kernel_ms"]*1000,2), "us (events)")'; grep checksum $OUT/err_$e.log | tail -1; done
(cd /tmp && JDET_ROI_FWD_EXACT=1 timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p -o t -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --steps 10 --warmup 3 > $OUT/p.log 2>&1)
python - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/p/**/*counter_collection.csv",recursive=True):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "merged" in r["Kernel_Name"]: d["merged"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({c: round(sum(x)/len(x)) for c,x in d["merged"].items()})
PY
timeout 600 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_s2anet.py -x -q 2>&1 | tail -2
