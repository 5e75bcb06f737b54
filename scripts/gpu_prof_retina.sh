R=$PWD; mkdir -p $R/gpurun_out/prof_retina
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_retina/trace -o t -- python $R/bench.py --workload retinanet_infer --steps 10 --warmup 4 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_retina/trace.log 2>&1
cd $R
f=$(find gpurun_out/prof_retina/trace -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f nms_scan_kernel 1 6 | cut -c1-200 | head -45
rm -rf gpurun_out/prof_retina/trace
