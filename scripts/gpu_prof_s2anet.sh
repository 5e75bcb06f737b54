R=$PWD; mkdir -p $R/gpurun_out/prof_s2anet
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_s2anet/trace -o t -- python $R/bench.py --workload s2anet_train --steps 10 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/prof_s2anet/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_s2anet/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6, "kernels", len(rows))
out = []
for r in rows[:45]:
    out.append("%6.2f%% %9.3f ms %7d calls  avg %9.1f us  %s" % (float(r["Percentage"]), float(r["TotalDurationNs"])/1e6, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
open("gpurun_out/prof_s2anet/summary.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
