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

run_final() {
# round 5 evidence run: smoke(), the full GPU suite, the default bench line (with `secondary`), the same command under
# rocprofv3 (steady-state step breakdown + roofline-kernel rows), traffic counters of the roofline kernel (one --pmc set
# per pass), kernel stats of the RoIAlign forward / backward, MFMA counters of the whole S2ANet step.
# Output: gpurun_out/r5_final/ (what is judged is copied into profiles/r05_*).
OUT=$R/gpurun_out/r5_final; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-2600
grep -c "AccumulateGrad" $OUT/bench_default.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/trace_default.log 2>&1)
f=$(find $OUT/trace_default -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f assign_anchor_kernel 4 5 120 > $OUT/steady_state_s2anet.txt 2>&1
head -3 $OUT/steady_state_s2anet.txt | cut -c1-160
k=$(find $OUT/trace_default -name '*kernel_stats.csv' | head -1)
head -1 $k > $OUT/roofline_kernel_stats.csv
grep "roi_align_fwd_merged_kernel\|roi_order_kernel\|conv_bn_kernel\|conv3x3_wgrad_kernel\|conv3x3_igemm_kernel\|bn_out_bwd_kernel\|bn_sums_finish_kernel\|dgrad_weights_kernel\|channel_sum_kernel" $k >> $OUT/roofline_kernel_stats.csv
cut -c1-240 $OUT/roofline_kernel_stats.csv
rm -rf $OUT/trace_default
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_READ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_fwd_$n -o p -- python $R/bench.py --workload roi_align_rotated --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $OUT/pmc_fwd_$n.log 2>&1)
done
python - <<PY > $OUT/roi_align_fwd_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_fwd_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if "roi_" in k:
        for c, v in sorted(cs.items()):
            print("%-70s %-32s mean %.6g over %d dispatches" % (k, c, sum(v) / len(v), len(v)))
PY
cut -c40-200 $OUT/roi_align_fwd_counters.txt
rm -rf $OUT/pmc_fwd_*/
for wl in roi_align_rotated roi_align_rotated_bwd; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o t -- python $R/bench.py --workload $wl --no-cpu-baseline --no-secondary > $OUT/trace_$wl.log 2>&1)
  k=$(find $OUT/trace_$wl -name '*kernel_stats.csv' | head -1)
  [ -n "$k" ] && head -12 $k | cut -c1-220 > $OUT/kernel_stats_$wl.csv
  grep -o '"ms_per_step": [0-9.]*' $OUT/trace_$wl.log | head -1
  rm -rf $OUT/trace_$wl
  echo "== $wl"; cut -c1-150 $OUT/kernel_stats_$wl.csv | head -7
done
for c in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 900 rocprofv3 --pmc $c -f csv -d $OUT/mfma_$n -o p -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/mfma_$n.log 2>&1 || echo "pmc $c failed")
done
python - <<PY > $OUT/s2anet_mfma_utilisation.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/mfma_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:90]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:25]
print("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); sums over the dispatches of 6 steps (3 warm-up + 3 timed)")
num = den = 0.0
for k, c in rows:
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    frac = busy / (act / 8 * 1024) if act else float("nan")
    num += busy; den += act / 8 * 1024
    print("%-90s calls %4d  mfma_busy %6.1f %%  MOPS_F32 %.3e" % (k, cnt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 100 * frac, c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)))
print("cycle-weighted MFMA busy over these kernels: %.1f %%" % (100 * num / den if den else float("nan")))
PY
tail -1 $OUT/s2anet_mfma_utilisation.txt
rm -rf $OUT/mfma_*/
}

run_i() {   # closing re-validation: two-rank tests, 16-deep K steps on the short reductions, the default bench line
  OUT=$R/gpurun_out/r5_i; mkdir -p $OUT
  timeout 900 python -m pytest tests/test_gpu_ddp_detectors.py tests/test_gpu_frozen_bn.py -q 2>&1 | tail -4 | tee $OUT/pytest.txt
  timeout 600 python scripts/conv_bn_timing.py tiles 2>&1 | grep -v Warning | tee $OUT/tiles.txt
  timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-400
  grep -c "AccumulateGrad\|Grad strides" $OUT/bench_default.err
}

run_j() {   # prediction convs through conv_module + 16-deep K steps: model suites, the bench line, the step profile
  OUT=$R/gpurun_out/r5_j; mkdir -p $OUT
  timeout 1200 python -m pytest tests/test_gpu_s2anet.py tests/test_gpu_head_parity.py tests/test_gpu_oriented_rcnn.py tests/test_gpu_roi_transformer.py tests/test_gpu_conv_bn.py tests/test_gpu_conv_igemm.py tests/test_gpu_configs_full_size.py tests/test_gpu_ddp_detectors.py -q 2>&1 | tail -4 | tee $OUT/pytest.txt
  timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-400
  bash scripts/gpu_prof_s2anet.sh > $OUT/prof.txt 2>&1; cp gpurun_out/prof_s2anet/steady_state.txt $OUT/steady_state.txt; head -3 $OUT/steady_state.txt | cut -c1-150
  rm -rf gpurun_out/prof_s2anet/trace
}

run_k() {   # backward with direct rows (csr_gather.h): parity, time, per-kernel times.  The A/B runs behind
            # profiles/r05_roi_bwd_notes.md (cap 0 / 16 / 64, counter atomics back to back, LDS merge, fma merge, XCD-sorted
            # RoIs) used switches that lived in the library for those runs only
  OUT=$R/gpurun_out/r5_k; mkdir -p $OUT
  timeout 600 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_reference_kernels.py -q -x -k "back or bwd or grad or workspace or cap or golden or channels_last or riroi" 2>&1 | tail -2 | tee $OUT/pytest.txt
  for i in 1 2 3; do
    timeout 300 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --steps 200 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done | tee $OUT/bwd.txt
  cd /tmp
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $R/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --steps 100 > /dev/null 2>&1
  k=$(find $OUT/trace -name '*kernel_stats.csv' | head -1); cp $k $OUT/kernel_stats_roi_align_rotated_bwd.csv; rm -rf $OUT/trace
  cd $R
}

run_t() {   # traffic counters of the backward's four kernels (direct rows), one --pmc set per pass
  OUT=$R/gpurun_out/r5_t; mkdir -p $OUT
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-30)
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python $R/bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $OUT/pmc_$n.log 2>&1)
  done
  python - <<PY > $OUT/roi_align_bwd_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if "csr_" in k or "bwd_patch" in k:
        for c, v in sorted(cs.items()):
            print("%-70s %-22s mean %.6g over %d dispatches" % (k, c, sum(v) / len(v), len(v)))
PY
  cat $OUT/roi_align_bwd_counters.txt | cut -c1-150
  rm -rf $OUT/pmc_*/
}

run_z() {   # closing validation of the final tree: smoke, the whole GPU suite, the default bench line
  OUT=$R/gpurun_out/r5_z; mkdir -p $OUT
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
  timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-300
  grep -c "AccumulateGrad\|Grad strides" $OUT/bench_default.err
  true
}

run=${1:-}; [ $# -gt 0 ] && shift
case "$run" in
  a|b|c|d|e|f|g|h|i|j|k|t|z|final) run_$run "$@";;
  *) echo "usage: gpu_r5.sh {a|b|c|d|e|f|g|h|i|j|z|final} [args]"; exit 2;;
esac
