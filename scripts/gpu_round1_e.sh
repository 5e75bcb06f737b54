python -m pytest tests/test_gpu_roi_align.py -m gpu -q -x 2>&1 | tail -2
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["roofline"]["kernel_ms"]*1e3,1),"us", round(d["roofline"]["achieved"]),"GB/s")'; }
for nw in 4 8 16; do echo "order NW=$nw SG=4: $(JDET_ROI_FWD_WAVES=$nw run)"; done
for sg in 2 8; do echo "order NW=4 SG=$sg: $(JDET_ROI_FWD_SG=$sg run)"; done
echo "order NW=8 SG=8: $(JDET_ROI_FWD_WAVES=8 JDET_ROI_FWD_SG=8 run)"
for nw in 4 8; do echo "noorder NW=$nw: $(JDET_BENCH_NO_ORDER=1 JDET_ROI_FWD_WAVES=$nw run)"; done
for abl in 1 2 3 4 7; do for nw in 4 8; do echo "noorder ABL=$abl NW=$nw: $(JDET_BENCH_NO_ORDER=1 JDET_ROI_ABLATE=$abl JDET_ROI_FWD_WAVES=$nw run)"; done; done
