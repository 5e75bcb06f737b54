#!/bin/bash
# Round 4: graph-mode divergence bisect (shared-pool / own-pool / eager update), backward A/B (scan folded or not),
# S2ANet with P4 packed with the small levels
set -u
R=$PWD
OUT=$R/gpurun_out/r4_g; mkdir -p $OUT
export TMPDIR=/tmp
for m in shared own eager; do
  JDET_GRAPH_UPDATE=$m timeout 600 python scripts/ddp_graph_diag.py orcnn 7 2 > $OUT/ddp_$m.log 2>&1
  echo "== update=$m"; grep -E "RESULT|DISAGREE|GARBAGE" $OUT/ddp_$m.log | cut -c1-330 | head -8
done
trace() {  # $1 = tag, $2 = workload, rest = env
  tag=$1; wl=$2; shift; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$tag -o t -- python $R/bench.py --workload $wl --no-cpu-baseline --steps 50 > $OUT/t_$tag.log 2>&1)
  python - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/t_$tag/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    m=re.search(r"((roi|csr|bwd|riroi)_\w+_kernel)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=sum(sum(v[5:])/len(v[5:]) for k,v in d.items() if len(v)>5)
print("[$tag] total %.1f :"%tot, "; ".join("%s %.1f"%(k,sum(v[5:])/len(v[5:])) for k,v in d.items() if len(v)>5))
PY
}
trace bwd_fold roi_align_rotated_bwd A=1
trace bwd_nofold roi_align_rotated_bwd JDET_ROI_BWD_FOLD_SCAN=0
timeout 300 python -m pytest tests/test_gpu_roi_align.py -x -q -k "channels_last or kept_workspace or cfg0 or full_size" 2>&1 | tail -2
for p in 1024 4096; do JDET_PACK_MAX_POS=$p timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("s2anet pack_max_pos='$p'", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms")'; done
