mkdir -p gpurun_out
python -m pytest tests/test_gpu_roi_align.py -m gpu -q -x > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c.log
for ord in 1 0; do for nw in 4 8; do for sg in 4; do
  echo "NOORDER=$ord NW=$nw SG=$sg: $(JDET_BENCH_NO_ORDER=$ord JDET_ROI_FWD_WAVES=$nw JDET_ROI_FWD_SG=$sg python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["roofline"]["kernel_ms"], d["roofline"]["achieved"])')"
done; done; done
JDET_ROI_FWD_WAVES=8 bash scripts/gpu_pmc.sh prof_fwd_v3 2>&1 | grep -E "FETCH|WRITE_SIZE|TCC_|roi_align_fwd|roi_order"
