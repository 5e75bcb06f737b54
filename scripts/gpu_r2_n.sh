#!/bin/bash
# round 2, call N: fixed-shape RoI-Transformer path (tests, step times), GEMM layout of the DeformConv forward, graph-mode bench
set -u
OUT=gpurun_out/r2_n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_transformer.py tests/test_gpu_oriented_rcnn.py tests/test_gpu_configs_full_size.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for wl in orcnn_train roitrans_r50_train roitrans_train; do
  timeout 400 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -1 $OUT/bench_$wl.json | cut -c1-200
  JDET_TRAIN_GRAPH=1 timeout 400 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_${wl}_graph.json 2> $OUT/bench_${wl}_graph.err
  tail -1 $OUT/bench_${wl}_graph.json | cut -c1-200; tail -2 $OUT/bench_${wl}_graph.err | cut -c1-300
done
JDET_TRAIN_GRAPH=1 timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_s2anet_graph.json 2> $OUT/bench_s2anet_graph.err
tail -1 $OUT/bench_s2anet_graph.json | cut -c1-200
timeout 200 python - <<'PY' 2>&1 | tee $OUT/mm_layout.txt
import torch, time
dev = "cuda"
M, N, K = 32768, 256, 2304
cols = torch.randn(M, K, device=dev); wt = torch.randn(N, K, device=dev); wkn = wt.t().contiguous()
g = torch.randn(M, N, device=dev)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("fwd cols @ wt.t() (NT)          %.1f us" % t(lambda: torch.mm(cols, wt.t())))
print("fwd cols @ wkn (NN)             %.1f us" % t(lambda: torch.mm(cols, wkn)))
print("fwd (wt @ cols.t()).t()         %.1f us" % t(lambda: torch.mm(wt, cols.t())))
print("fwd F.linear(cols, wt)          %.1f us" % t(lambda: torch.nn.functional.linear(cols, wt)))
out = torch.empty(M, N, device=dev)
print("fwd mm out= NN                  %.1f us" % t(lambda: torch.mm(cols, wkn, out=out)))
print("bwd g @ wt (NN)                 %.1f us" % t(lambda: torch.mm(g, wt)))
print("wrw g.t() @ cols (TN)           %.1f us" % t(lambda: torch.mm(g.t(), cols)))
for M2 in (8192, 2048, 512, 128):
    c2 = cols[:M2]
    print("M=%d NT %.1f  NN %.1f us" % (M2, t(lambda: torch.mm(c2, wt.t())), t(lambda: torch.mm(c2, wkn))))
PY
