#!/bin/bash
# round 2, call P: RiRoIAlign on the vector forward + mixed-row gather backward: parity, timings
set -u
OUT=$PWD/gpurun_out/r2_p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_roi_align.py tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python - <<'PY' 2>&1 | tee $OUT/riroi_timing.txt
import time, torch, numpy as np
from tests import inputs as I
from jdet_amd.ops.riroi_align import RiRoIAlign
from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
dev = "cuda"
rng = np.random.default_rng(0)
x = torch.randn(1, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, 2000, extent=1024.0, wh=(8.0, 256.0)), np.zeros(2000))).to(dev)
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, layer in (("RiRoIAlign nO=8", RiRoIAlign((7, 7), 0.25, 2, 8)), ("ROIAlignRotated", ROIAlignRotated((7, 7), 0.25, 2))):
    with torch.no_grad():
        fwd = t(lambda: layer(x, rois))
    y = layer(x, rois)
    g = torch.randn_like(y)
    def fb():
        x.grad = None
        layer(x, rois).backward(g)
    print("%-18s forward %.1f us   forward+backward %.1f us" % (name, fwd, t(fb)))
PY
