for dbg in 0 1 2 3 4 5 7; do for nw in 4 8; do
  echo "DBG=$dbg NW=$nw: $(JDET_BENCH_NO_ORDER=1 JDET_ROI_DBG=$dbg JDET_ROI_FWD_WAVES=$nw python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["roofline"]["kernel_ms"])')"
done; done
