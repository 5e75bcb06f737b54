python -m pytest tests/test_gpu_roi_align.py -m gpu -q -x 2>&1 | tail -5
run() { python bench.py --workload roi_align_rotated_bwd --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["roofline"]["kernel_ms"]*1e3,1),"us", round(d["roofline"]["achieved"]),"GB/s")'; }
echo "bwd gather: $(run)"
echo "bwd atomic: $(JDET_BENCH_BWD_ATOMIC=1 run)"
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_bwd -o t -- python $R/bench.py --workload roi_align_rotated_bwd --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_bwd/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%6.2f%% avg %9.1f us %5s calls %s" % (float(r["Percentage"]), float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:90]))
PY
