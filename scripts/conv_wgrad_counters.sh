#!/bin/bash
# usage (GPU box): bash scripts/conv_wgrad_counters.sh [ksplit ...] -- SQ counters of csrc/conv_wgrad.hip (per forced
# split) and of the library's weight gradient at the 2 x 128^2 x 256 problem (one --pmc set per pass), averages per
# kernel -> gpurun_out/wgrad_pmc/summary_<ks>.txt
set -u
R=$PWD; OUT=$R/gpurun_out/wgrad_pmc; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/wgrad_loop.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from jdet_amd.ops import conv_igemm as CI
ks = int(sys.argv[1]); hw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
x = torch.randn(2, 256, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
g = torch.randn(2, 256, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
out = torch.zeros(256, 3, 3, 256, device="cuda")
for _ in range(12):
    CI.conv3x3_wgrad_nhwc(x.permute(0, 2, 3, 1), g.permute(0, 2, 3, 1), out=out, ksplit=ks)
    if ks == 0:
        torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
torch.cuda.synchronize()
PY
cd /tmp
for ks in "${@:-0}"; do
  i=0
  for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/k${ks}_p$i -o t -- python /tmp/wgrad_loop.py $ks ${HW:-128} > $OUT/k${ks}_p$i.log 2>&1
  done
  python - <<PY > $OUT/summary_$ks.txt
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/k${ks}_p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wgrad" in k or "wrw" in k:
            d[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-34s %16.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
  echo "== ksplit $ks"; cat $OUT/summary_$ks.txt
done
