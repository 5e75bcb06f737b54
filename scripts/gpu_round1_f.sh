mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py > gpurun_out/bench_r1.log 2>&1; tail -1 gpurun_out/bench_r1.log
JDET_ROI_FWD_WAVES=8 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
bash scripts/gpu_pmc.sh prof_fwd_v4 > gpurun_out/prof_fwd_v4.txt 2>&1; grep -E "roi_order|roi_align_fwd" gpurun_out/prof_fwd_v4.txt | head -8
