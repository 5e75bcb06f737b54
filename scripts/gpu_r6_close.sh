#!/bin/bash
# round 6 closing measurements (GPU box): usage  bash scripts/gpu_r6_close.sh <step> [...]
#   mfma   MFMA counters of the whole S2ANet step (the recipe of gpu_r5.sh run_final) -> gpurun_out/r6_mfma.txt
#   ab     same-box A/B of the head towers' / FPN's weight and data gradients on the own kernels vs the library's
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
run_mfma() {
OUT=$R/gpurun_out/r6_mfma_tmp; mkdir -p $OUT
for c in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 900 rocprofv3 --pmc $c -f csv -d $OUT/mfma_$n -o p -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/mfma_$n.log 2>&1 || echo "pmc $c failed")
done
python - <<PY > $R/gpurun_out/r6_mfma.txt
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
tail -1 $R/gpurun_out/r6_mfma.txt
rm -rf $OUT
}
run_ab() {
# (run BEFORE run_mfma on a fresh box: the library's solver search under the counters' serialised timing writes its picks to the
#  user find-db, and every later process of the box then runs those -- 28.4 instead of 26.5 ms in the first attempt)
bash scripts/ab_step.sh -n 2 "JDET_CONV_WGRAD=0" "JDET_CONV_WGRAD=1" "JDET_CONV_IGEMM_DGRAD=1" "JDET_BENCH_FORCE_DIST=1" 2>&1 | tee $R/gpurun_out/r6_ab_own_grads_deep.txt
}
run_ab2() {   # weight gradients of the fused bottlenecks on a side stream; the towers' own weight gradient after the spill fix; DDP on one GPU
timeout 900 python -m pytest tests/test_gpu_conv_wgrad.py tests/test_gpu_conv_bn.py -q -x 2>&1 | tail -3
JDET_BOTTLENECK_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_gpu_conv_bn.py tests/test_gpu_s2anet.py -q -x 2>&1 | tail -3
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
bash scripts/ab_step.sh -n 2 "JDET_BOTTLENECK_WGRAD_STREAM=0" "JDET_BOTTLENECK_WGRAD_STREAM=1" "JDET_CONV_WGRAD=1" "JDET_BENCH_FORCE_DIST=1" "JDET_BENCH_FORCE_DIST=1 JDET_BOTTLENECK_WGRAD_STREAM=1" 2>&1 | tee $R/gpurun_out/r6_ab_wgrad_stream.txt
}
run_k16() {   # 16-deep K steps under the two-steps-ahead operand requests: per layer, then the step
for k in 0 1 2; do echo "== JDET_CONV_BN_K16=$k"; JDET_CONV_BN_K16=$k timeout 600 python scripts/conv_bn_timing.py layers 2>&1 | grep -v Warn; done | tee $R/gpurun_out/r6_conv_bn_k16_layers.txt | grep "==\|sum"
bash scripts/ab_step.sh -n 2 "JDET_CONV_BN_K16=0" "JDET_CONV_BN_K16=1" "JDET_CONV_BN_K16=2" 2>&1 | tee $R/gpurun_out/r6_ab_k16.txt
}
run_ab3() {   # the towers' weight gradient on the own kernel's 128 x 64 tile (new rule) / data gradient on the own kernel
timeout 900 python -m pytest tests/test_gpu_conv_wgrad.py tests/test_gpu_conv_igemm.py -q -x 2>&1 | tail -3
bash scripts/ab_step.sh -n 3 "JDET_CONV_WGRAD=0" "JDET_CONV_WGRAD=1" "JDET_CONV_WGRAD=1 JDET_CONV_IGEMM_DGRAD=1" 2>&1 | tee $R/gpurun_out/r6_ab_wgrad_mid.txt
JDET_CONV_WGRAD=1 bash scripts/gpu_prof_s2anet.sh > /dev/null 2>&1; cp gpurun_out/prof_s2anet/steady_state.txt gpurun_out/r6_steady_wgrad_own.txt; head -3 gpurun_out/r6_steady_wgrad_own.txt | cut -c1-150; rm -rf gpurun_out/prof_s2anet/trace
}
run_ab4() {   # the towers' data gradient on the own kernel with the flipped weights from one bank launch
timeout 900 python -m pytest tests/test_gpu_conv_igemm.py -q -x 2>&1 | tail -3
bash scripts/ab_step.sh -n 3 "JDET_CONV_WGRAD=0" "JDET_CONV_WGRAD=1" "JDET_CONV_WGRAD=1 JDET_CONV_IGEMM_DGRAD=1" "JDET_CONV_WGRAD=0 JDET_CONV_IGEMM_DGRAD=1" 2>&1 | tee $R/gpurun_out/r6_ab_dgrad_bank.txt
JDET_CONV_WGRAD=1 JDET_CONV_IGEMM_DGRAD=1 bash scripts/gpu_prof_s2anet.sh > /dev/null 2>&1; cp gpurun_out/prof_s2anet/steady_state.txt gpurun_out/r6_steady_own_grads.txt; head -3 gpurun_out/r6_steady_own_grads.txt | cut -c1-150; rm -rf gpurun_out/prof_s2anet/trace
}
run_abl() {   # ablation builds of the conv_bn 64-tile K loop (timing only): 1 no requests / LDS stores, 2 no MFMAs, 4 no barriers
for k in ${ABLS:-0 1 2 3 4 5 6}; do echo "== JDET_CONV_BN_ABL=$k"; JDET_CONV_BN_ABL=$k timeout 600 python scripts/conv_bn_timing.py layers 2>&1 | grep -v Warn | cut -c1-42; done | tee $R/gpurun_out/r6_conv_bn_abl_layers.txt | grep "==\|sum\|l3.conv2\|l2.conv2\|l1.conv2\|l3.conv1"
}
for s in "$@"; do run_$s; done
