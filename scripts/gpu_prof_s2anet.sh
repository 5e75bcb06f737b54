R=$PWD; mkdir -p $R/gpurun_out/prof_s2anet
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_s2anet/trace -o t -- python $R/bench.py --workload s2anet_train --steps 8 --warmup 4 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/prof_s2anet/trace.log 2>&1
cd $R
f=$(find gpurun_out/prof_s2anet/trace -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 5 | tee gpurun_out/prof_s2anet/steady_state.txt
tail -2 gpurun_out/prof_s2anet/trace.log | cut -c1-300
