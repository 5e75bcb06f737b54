import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ["JDET_ROI_FWD_GRAN"] = "1024"; os.environ["JDET_ROI_SPLIT_SKIP_LOOP"] = "1"
import r6_ring_check as RC
from jdet_amd import _lib as L
K = int(os.environ.get("CASE", "2"))
lib = L.lib(); dev = torch.device("cuda:0")
variant, no, feat, rois, hw, scale = RC.cases()[K]
x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
r = torch.from_numpy(rois).to(dev)
N, C, H, W = x.shape; R = r.shape[0]
out = torch.full((R, C) + tuple(hw), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
wsb = lib.jdet_roi_align_forward_cl_workspace(R, hw[0], hw[1]); ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev)
ws[:] = 0xAB
L.check(lib.jdet_roi_align_forward_cl(variant, x.data_ptr(), N, C, H, W, r.data_ptr(), R, hw[0], hw[1], scale, 2, no, out.data_ptr(), ws.data_ptr(), wsb, L.stream_ptr(x)), "fwd")
torch.cuda.synchronize()
off = (256 + 8 * R + 255) // 256 * 256
plan = ws[off:off + R * 4 * 640 * 4].view(torch.int32).view(R, 4, 640).cpu().numpy()
hdr = plan[:, :, 576:582]
print("R", R, "nbins", hw, "gtot min/max", hdr[..., 0].min(), hdr[..., 0].max(), "split range", hdr[..., 1].min(), hdr[..., 1].max(), "batch", np.unique(hdr[..., 2]))
bad = np.argwhere((hdr[..., 0] < 0) | (hdr[..., 0] > 64) | (hdr[..., 1] < 0) | (hdr[..., 1] > hdr[..., 0]))
print("bad headers:", bad[:10], "masked rois:", int((rois[:, 0] < 0).sum()))
for (ri, wv) in bad[:5]:
    print(ri, wv, hdr[ri, wv], rois[ri])
