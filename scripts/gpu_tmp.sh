bash scripts/gpu_pmc.sh r01_fwd_merged_nt --workload roi_align_rotated > gpurun_out/pmc_merged_nt.log 2>&1
grep "FETCH_SIZE\|WRITE_SIZE\|TCC_HIT\|TCC_MISS\|roi_align_fwd_merged_kernel<0, 4, 0>(float const\*, float const\*, float\*, int, int, int, int, int, float, int const\*, int), " gpurun_out/pmc_merged_nt.log | cut -c1-200
rm -rf gpurun_out/r01_fwd_merged_nt/trace gpurun_out/r01_fwd_merged_nt/pmc_*/ 2>/dev/null
timeout 900 python bench.py --workload roitrans_r50_train --batch 2 --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
