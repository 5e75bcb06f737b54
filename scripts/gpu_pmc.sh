# usage: bash scripts/gpu_pmc.sh <tag> [bench args...]   -- rocprofv3 kernel trace + separate PMC passes
tag=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/$tag/trace -o t -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 5 "$@" > $R/gpurun_out/$tag/trace.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -f csv -d $R/gpurun_out/$tag/pmc_$n -o p -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 "$@" > $R/gpurun_out/$tag/pmc_$n.log 2>&1 || echo "pmc $c failed"
done
cd $R
python scripts/summarize_prof.py gpurun_out/$tag
