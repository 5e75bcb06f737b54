run() { timeout 300 python bench.py --workload roi_align_rotated --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"]*1000,1),"us/step")'; }
echo "default: $(run)"
echo "abl16 float trig: $(JDET_ROI_ABLATE=16 run)"
echo "default again: $(run)"
