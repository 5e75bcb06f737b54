#!/bin/bash
# usage (GPU box): bash scripts/conv_igemm_counters.sh -- SQ counters of the conv igemm kernel and of the library conv at
# the 2 x 128^2 x 256 problem (one --pmc set per pass), averages per kernel -> gpurun_out/conv_pmc/summary.txt
set -u
R=$PWD; OUT=$R/gpurun_out/conv_pmc; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/conv_loop.py <<PY
import sys, torch, torch.nn.functional as F
sys.path.insert(0, "$R")
from jdet_amd.ops import conv_igemm as CI
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(2, 256, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
b = torch.randn(256, device="cuda")
xn, wk = x.permute(0, 2, 3, 1), CI.weight_krsc(w)
for _ in range(20):
    CI.conv3x3_nhwc(xn, wk, b, True)
    F.conv2d(x, w, None, padding=1)
torch.cuda.synchronize()
PY
i=0
cd /tmp
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o t -- python /tmp/conv_loop.py ${1:-128} > $OUT/p$i.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm" in k or "conv" in k.lower():
            d[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-34s %16.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
cat $OUT/summary.txt
