#!/usr/bin/env python3
"""Phase timeline of the tile-stationary RoIAlign forward (jdet_debug_roi_tile_timeline): per-workgroup s_memtime
stamps -> where a workgroup's time goes, how many workgroups share a CU, how long the whole grid runs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jdet_amd import _lib as L  # noqa: E402
from tests import inputs as I  # noqa: E402

dev = torch.device("cuda:0")
R = 2000
rng = np.random.default_rng(1000)
g = torch.Generator(device="cpu").manual_seed(1000)
feat = torch.randn((1, 256, 256, 256), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))).to(dev)
out = torch.empty((R, 256, 7, 7), device=dev, memory_format=torch.channels_last)
lib = L.lib()
wsb = lib.jdet_roi_align_forward_cl_workspace(1, 256, 256, R, 7, 7)
ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev)


def run():
    L.check(lib.jdet_roi_align_forward_cl(0, feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), R, 7, 7, 0.25, 2, 0,
                                          out.data_ptr(), ws.data_ptr(), wsb, L.stream_ptr(feat)), "fwd")


for _ in range(5):
    run()
torch.cuda.synchronize()
nblk = 8192
buf = torch.zeros((nblk, 32), dtype=torch.int64, device=dev)
lib.jdet_debug_roi_tile_timeline(buf.data_ptr())
run()
torch.cuda.synchronize()
lib.jdet_debug_roi_tile_timeline(None)
t = buf.cpu().numpy().astype(np.uint64)
pl = t[4096:]
pl = pl[pl[:, 0] > 0]
print("plan workgroups:", len(pl))
for i, nm in enumerate(["scan", "records (trig)", "ownership+prefix", "alloc", "tables written"]):
    d = (pl[:, i + 1] - pl[:, i]).astype(np.float64)
    print("  plan %-18s mean %8.0f  p90 %8.0f  max %8.0f" % (nm, d.mean(), np.percentile(d, 90), d.max()))
d = (pl[:, 5] - pl[:, 0]).astype(np.float64)
print("  plan workgroup lifetime  mean %8.0f  max %8.0f" % (d.mean(), d.max()))
t = t[:4096]
live = t[:, 0] > 0
t = t[live]
print("workgroups that ran:", live.sum())
t0 = t[:, 0].min()
start = (t[:, 0] - t0).astype(np.float64)
end = (t[:, 29] - t0).astype(np.float64)
print("grid span (s_memtime ticks): %.0f ; wg lifetime mean %.0f  min %.0f max %.0f" % (end.max(), (end - start).mean(), (end - start).min(), (end - start).max()))
print("start times: p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(start, [50, 90, 100])))
t = t[t[:, 29] > 0]
start = (t[:, 0] - t0).astype(np.float64)
end = (t[:, 29] - t0).astype(np.float64)
print("pool workgroups with work:", len(t), " lifetime mean %.0f max %.0f" % ((end - start).mean(), (end - start).max()))
d = (t[:, 1] - t[:, 0]).astype(np.float64)
print("prologue (prefix, share)  mean %8.0f  p90 %8.0f" % (d.mean(), np.percentile(d, 90)))
nseg = (t[:, 30] >> np.uint64(32)).astype(np.int64)
nbins = (t[:, 30] & np.uint64(0xFFFFFFFF)).astype(np.int64)
print("share: bins mean %.1f max %d ; tile segments per share mean %.2f max %d" % (nbins.mean(), nbins.max(), nseg.mean(), nseg.max()))
print("(chunk stamps below are those of the LAST segment of a share)")
has = t[:, 5] > 0
for cc in range(8):
    a, b, c = 4 + 2 * cc, 5 + 2 * cc, 6 + 2 * cc
    if not (t[has][:, c] > 0).all():
        break
    w = (t[has][:, b] - t[has][:, a]).astype(np.float64)   # store_window + wait for the prefetch + barrier
    k = (t[has][:, c] - t[has][:, b]).astype(np.float64)   # prefetch issue + compute
    print("chunk %d: wait+store %7.0f (p90 %7.0f)   compute %7.0f (p90 %7.0f)" % (cc, w.mean(), np.percentile(w, 90), k.mean(), np.percentile(k, 90)))
hw = t[:, 31]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
# per CU (one clock domain for sure): span and slot utilisation
spans, utils, cnts = [], [], []
for k_ in np.unique(key):
    m = key == k_
    st_, en_ = t[m][:, 0].astype(np.float64), t[m][:, 29].astype(np.float64)
    sp = en_.max() - st_.min()
    spans.append(sp); utils.append((en_ - st_).sum() / (2 * sp)); cnts.append(m.sum())
spans, utils, cnts = np.array(spans), np.array(utils), np.array(cnts)
print("per-CU span: mean %.0f p10 %.0f p90 %.0f max %.0f ticks ; slot utilisation mean %.2f ; wgs/CU min %d max %d"
      % (spans.mean(), np.percentile(spans, 10), np.percentile(spans, 90), spans.max(), utils.mean(), cnts.min(), cnts.max()))
print("distinct CUs used:", len(np.unique(key)), " workgroups per CU: mean %.2f max %d" % (len(key) / len(np.unique(key)), np.bincount(key).max()))
# concurrency: for each CU, max number of workgroups alive at the same time
mx = 0
for k_ in np.unique(key):
    m = key == k_
    ev = sorted([(s, 1) for s in start[m]] + [(e, -1) for e in end[m]])
    c = 0
    for _, d_ in ev:
        c += d_
        mx = max(mx, c)
print("max concurrently resident workgroups on one CU:", mx)
