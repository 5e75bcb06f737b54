run() { timeout 900 python bench.py --workload s2anet_train --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2),"ms", round(d["value"],2),"img/s")'; }
echo "default: $(run)"
echo "cudnn.benchmark: $(JDET_CUDNN_BENCHMARK=1 run)"
echo "MIOPEN_FIND_MODE=NORMAL: $(MIOPEN_FIND_MODE=NORMAL run)"
echo "MIOPEN_FIND_ENFORCE=SEARCH + benchmark: $(MIOPEN_FIND_ENFORCE=3 JDET_CUDNN_BENCHMARK=1 timeout 1200 python bench.py --workload s2anet_train --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200)"
