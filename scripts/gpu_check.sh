#!/bin/bash
# quick GPU check: a list of test files + bench lines of the given workloads
# usage: gpu_check.sh "<pytest args>" [workload ...]
set -u
OUT=$PWD/gpurun_out/check; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest $1 -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
shift
for wl in "$@"; do
  timeout 400 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  python - <<PY
import json
try:
    l = json.loads(open("$OUT/bench_$wl.json").read().strip().splitlines()[-1])
    print("$wl: %.3f ms/step  value %.2f %s" % (l["ms_per_step"], l["value"], l["unit"]))
except Exception as e:
    print("$wl FAILED", e)
PY
done
