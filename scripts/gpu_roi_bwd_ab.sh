timeout 600 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_dcn_arf.py tests/test_gpu_oriented_rcnn.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --workload roi_align_rotated_bwd --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"]*1000,1),"us/step")'
