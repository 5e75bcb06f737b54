#!/bin/bash
# usage (GPU box): bash scripts/gpu_r5.sh <run> [args]   -- the round-5 measurement runs, one function per run
set -u
R=$PWD
export TMPDIR=/tmp

run_a() {   # first contact of the bottleneck family: parity, per-layer / per-block times, step A/B
  OUT=$R/gpurun_out/r5_a; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py layers blocks wgrad tiles 2>&1 | grep -v Warning | tee $OUT/timing.txt
  bash scripts/ab_step.sh -n 1 -s 20 "JDET_BOTTLENECK_FUSED=1" "JDET_BOTTLENECK_FUSED=0" 2>&1 | tee $OUT/ab.txt
}

run_b() {   # after the tile rule + library weight gradients: the whole parity file, block times, step A/B, step profile
  OUT=$R/gpurun_out/r5_b; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py -q 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py blocks layers 2>&1 | grep -v Warning | tee $OUT/timing.txt
  JDET_BOTTLENECK_WGRAD=own timeout 300 python scripts/conv_bn_timing.py blocks 2>&1 | grep -v Warning | tee $OUT/timing_own_wgrad.txt
  bash scripts/ab_step.sh -n 2 -s 20 "JDET_BOTTLENECK_FUSED=1" "JDET_BOTTLENECK_FUSED=0" 2>&1 | tee $OUT/ab.txt
  bash scripts/gpu_prof_s2anet.sh > $OUT/prof.txt 2>&1; cp gpurun_out/prof_s2anet/steady_state.txt $OUT/steady_state_fused.txt
  rm -rf gpurun_out/prof_s2anet/trace
}

run_c() {   # epilogue prefetch + side-stream aliases: parity, two-rank tests, times, DDP overhead on one GPU
  OUT=$R/gpurun_out/r5_c; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py tests/test_gpu_ddp_detectors.py tests/test_gpu_s2anet.py tests/test_gpu_frozen_bn.py tests/test_gpu_conv1x1.py -q 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py layers blocks 2>&1 | grep -v Warning | tee $OUT/timing.txt
  bash scripts/ab_step.sh -n 2 -s 20 "JDET_BOTTLENECK_FUSED=1" "JDET_BOTTLENECK_FUSED=0" 2>&1 | tee $OUT/ab.txt
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 JDET_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 8 > $OUT/force_dist.json 2> $OUT/force_dist.err
  grep -o '"ms_per_step": [0-9.]*' $OUT/force_dist.json; grep -c "AccumulateGrad" $OUT/force_dist.err
}

run_d() {   # the side-stream two-rank test with its full report + the one-GPU DDP overhead
  OUT=$R/gpurun_out/r5_d; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_ddp_detectors.py -q -x --tb=short -k "side_stream or eager" > $OUT/pytest_full.txt 2>&1; tail -5 $OUT/pytest_full.txt
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 JDET_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 8 > $OUT/force_dist.json 2> $OUT/force_dist.err
  grep -o '"ms_per_step": [0-9.]*' $OUT/force_dist.json; grep -c "AccumulateGrad" $OUT/force_dist.err
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 8 2> $OUT/plain.err | grep -o '"ms_per_step": [0-9.]*'; grep -c "AccumulateGrad" $OUT/plain.err
}

run_e() {   # backward gather: 2-D XCD blocks vs stripes (A/B), parity under the new map; weight-gradient tile variants
  OUT=$R/gpurun_out/r5_e; mkdir -p $OUT
  for m in 0 1 0 1; do
    JDET_GATHER_MAP=$m timeout 300 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --steps 200 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/map $m /"
  done | tee $OUT/bwd_ab.txt
  JDET_GATHER_MAP=1 timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_reference_kernels.py -q -k "back or bwd or grad" 2>&1 | tail -5 | tee $OUT/pytest_bwd.txt
  cd /tmp
  for m in 0 1; do
    JDET_GATHER_MAP=$m rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace$m -o t -- python $R/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --steps 100 > /dev/null 2>&1
    k=$(find $OUT/trace$m -name '*kernel_stats.csv' | head -1); grep "csr_\|bwd_patch" $k | cut -d, -f1-4 | cut -c1-160 | sed "s/^/map $m /"
    rm -rf $OUT/trace$m
  done | tee $OUT/bwd_kernels.txt
  cd $R
  timeout 600 python scripts/conv_bn_timing.py wgrad 2>&1 | grep -v Warning | tee $OUT/wgrad.txt
}

run_f() {   # own weight gradients (64 x 64 tiles) by default: parity, variants, block times, step A/B
  OUT=$R/gpurun_out/r5_f; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py tests/test_gpu_conv_wgrad.py tests/test_gpu_roi_align.py -q 2>&1 | tail -8 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py wgrad blocks 2>&1 | grep -v Warning | tee $OUT/timing.txt
  bash scripts/ab_step.sh -n 2 -s 20 "JDET_BOTTLENECK_WGRAD=own" "JDET_BOTTLENECK_WGRAD=lib" "JDET_BOTTLENECK_FUSED=0" 2>&1 | tee $OUT/ab.txt
}

run_g() {   # weight-gradient split in whole XCD rounds: parity, auto vs library, block times, step A/B
  OUT=$R/gpurun_out/r5_g; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py -q 2>&1 | tail -8 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py wgrad blocks 2>&1 | grep -v Warning | cut -c1-60 | tee $OUT/timing.txt
  bash scripts/ab_step.sh -n 2 -s 20 "JDET_BOTTLENECK_WGRAD=own" "JDET_BOTTLENECK_WGRAD=lib" "JDET_BOTTLENECK_FUSED=0" 2>&1 | tee $OUT/ab.txt
}

run_h() {   # any-C channel sum, head weight gradients through the 64-tile kernel (A/B), two-rank graph-mode diagnosis
  OUT=$R/gpurun_out/r5_h; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_conv_bn.py tests/test_gpu_frozen_bn.py tests/test_gpu_conv_igemm.py tests/test_gpu_conv_wgrad.py -q 2>&1 | tail -6 | tee $OUT/pytest.txt
  bash scripts/ab_step.sh -n 2 -s 20 "JDET_CONV_WGRAD=0" "JDET_CONV_WGRAD=1 JDET_CONV_WGRAD_GENERAL=1" "JDET_CONV_WGRAD=1" 2>&1 | tee $OUT/ab.txt
  timeout 1200 python scripts/ddp_graph_diag.py orcnn 12 ${1:-12} 2>&1 | grep "RESULT\|GARBAGE\|== run\|Error\|error" | tee $OUT/diag_own.txt
}

run=${1:-}; [ $# -gt 0 ] && shift
case "$run" in
  a|b|c|d|e|f|g|h) run_$run "$@";;
  *) echo "usage: gpu_r5.sh {a|b|c|d|e|f|g|h} [args]"; exit 2;;
esac
