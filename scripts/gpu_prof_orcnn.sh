R=$PWD; mkdir -p $R/gpurun_out/prof_orcnn
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_orcnn/trace -o t -- python $R/bench.py --workload orcnn_train --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_orcnn/trace.log 2>&1
cd $R
f=$(find gpurun_out/prof_orcnn/trace -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 4 | cut -c1-200 | head -42
tail -1 gpurun_out/prof_orcnn/trace.log | cut -c1-200
rm -rf gpurun_out/prof_orcnn/trace
