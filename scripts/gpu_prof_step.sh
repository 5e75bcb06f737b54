#!/bin/bash
# steady-state kernel breakdown of a train workload: gpu_prof_step.sh <workload> <marker kernel> <marker launches per step>
set -u
WL=$1; MARK=$2; PER=$3
R=$PWD; OUT=$R/gpurun_out/prof_$WL; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $R/bench.py --workload $WL --steps 8 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python scripts/steady_state.py $f $MARK $PER 4 120 | cut -c1-220 > $OUT/steady_state.txt
head -3 $OUT/steady_state.txt
grep -v "igemm_\|Cijk_\|ck16tensor\|ck::tensor" $OUT/steady_state.txt | head -70
tail -1 $OUT/trace.log | cut -c1-200
rm -rf $OUT/trace
