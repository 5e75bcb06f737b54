mkdir -p gpurun_out
R=$PWD
python -m pytest tests/test_gpu_roi_align.py -m gpu -q -x > gpurun_out/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_b.log
for nw in 4 8; do for sg in 1 4 8; do
  echo "NW=$nw SG=$sg: $(JDET_ROI_FWD_WAVES=$nw JDET_ROI_FWD_SG=$sg python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["roofline"]["kernel_ms"], d["roofline"]["achieved"])')"
done; done
