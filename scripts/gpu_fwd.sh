run() { timeout 300 python bench.py --workload roi_align_rotated --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"]*1000,1),"us/step")'; }
timeout 600 python -m pytest tests/test_gpu_roi_align.py -m gpu -x -q 2>&1 | tail -2
echo "default: $(run)"
echo "NW8: $(JDET_ROI_FWD_WAVES=8 run)"
