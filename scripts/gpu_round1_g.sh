python -m pytest tests/test_gpu_roi_align.py -m gpu -q -x 2>&1 | tail -5
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["roofline"]["kernel_ms"]*1e3,1),"us", round(d["roofline"]["achieved"]),"GB/s")'; }
echo "cache=1 order: $(run)"
echo "cache=1 noorder: $(JDET_BENCH_NO_ORDER=1 run)"
echo "cache=0 order: $(JDET_ROI_FWD_CACHE=0 run)"
