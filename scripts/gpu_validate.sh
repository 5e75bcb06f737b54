set -x
R=$PWD
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
# the driver's command, default flags: one JSON line
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-1800
# same command under rocprofv3 (kernel trace + stats): steady-state step breakdown + the roofline kernel's average
mkdir -p $R/gpurun_out/prof_default; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_default/trace -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_default/trace.log 2>&1
cd $R
f=$(find gpurun_out/prof_default/trace -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 5 > gpurun_out/prof_default/steady_state.txt
head -3 gpurun_out/prof_default/steady_state.txt | cut -c1-200
k=$(find gpurun_out/prof_default/trace -name '*kernel_stats.csv' | head -1)
head -1 $k > gpurun_out/prof_default/roofline_kernel_stats.csv
grep "roi_align_fwd_merged_kernel\|roi_order_kernel" $k >> gpurun_out/prof_default/roofline_kernel_stats.csv
cat gpurun_out/prof_default/roofline_kernel_stats.csv | cut -c1-300
rm -rf gpurun_out/prof_default/trace
