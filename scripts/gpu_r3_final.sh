#!/bin/bash
# usage (GPU box): bash scripts/gpu_r3_final.sh -- the round-3 closing run: smoke(), the full GPU suite, the default bench
# line, the same command under rocprofv3 (roofline-kernel rows + steady-state step breakdown), the other model workloads
set -x
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
cd $R
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 5 120 > $O/steady_state.txt; head -2 $O/steady_state.txt | cut -c1-160
k=$(find $O/trace -name '*kernel_stats.csv' | head -1)
head -1 $k > $O/roofline_kernel_stats.csv
grep "roi_align_fwd_merged_kernel\|roi_order_kernel\|conv3x3_igemm_kernel\|bias_act_bwd_kernel" $k >> $O/roofline_kernel_stats.csv
cut -c1-220 $O/roofline_kernel_stats.csv
rm -rf $O/trace
for w in retinanet_infer orcnn_train roitrans_r50_train roitrans_train; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-secondary > $O/bench_$w.json 2> $O/bench_$w.err
  tail -1 $O/bench_$w.json | cut -c1-260
done
