import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from tests import inputs as I
from oracle import ref_hip as RH
from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
from jdet_amd.ops import _roi_common as RC
dev=torch.device('cuda:0')
for R in (512,2000):
    rng = np.random.default_rng(R)
    feat = torch.from_numpy(rng.standard_normal((1, 256, 256, 256)).astype(np.float32)).to(dev)
    rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))).to(dev)
    ref_y = RH.roi_align_forward("rot", feat, rois, (7, 7), 0.25, 2)
    layer = ROIAlignRotated(7, 0.25, 2)
    x = feat.contiguous(memory_format=torch.channels_last)
    ys={}
    for mode in ("reference","merged"):
        prev=RC.set_arithmetic(mode); ys[mode]=layer(x,rois); RC.set_arithmetic(prev)
    print(R, "twin==refkernel %.4f  merged==refkernel %.4f  merged==twin %.4f  max|merged-twin| %.3e  max|merged-refk| %.3e max|twin-refk| %.3e" % (
        float((ys["reference"]==ref_y).float().mean()), float((ys["merged"]==ref_y).float().mean()), float((ys["merged"]==ys["reference"]).float().mean()),
        float((ys["merged"]-ys["reference"]).abs().max()), float((ys["merged"]-ref_y).abs().max()), float((ys["reference"]-ref_y).abs().max())))
