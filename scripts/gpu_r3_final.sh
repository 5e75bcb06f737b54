#!/bin/bash
# usage (GPU box): bash scripts/gpu_r3_final.sh -- full GPU suite, smoke, and the bench lines of every model workload
set -x
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
mkdir -p gpurun_out/final
timeout 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; tail -1 gpurun_out/final/bench_default.json | cut -c1-400
for w in retinanet_infer orcnn_train roitrans_r50_train roitrans_train; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err
  tail -1 gpurun_out/final/bench_$w.json | cut -c1-260
done
