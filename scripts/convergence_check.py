"""Sanity run on the GPU box: every train config overfits one fixed synthetic batch (loss falls, stays finite)."""
import sys, torch
sys.path.insert(0, ".")
import jdet_amd.models  # noqa
from jdet_amd.config.named import S2ANET_CFG, ORCNN_CFG, roitrans_train_cfg
from jdet_amd.runner import Runner, synthetic_batch
dev = torch.device("cuda:0")
for name, cfg in (("s2anet", S2ANET_CFG), ("orcnn", ORCNN_CFG), ("roitrans", roitrans_train_cfg())):
    torch.manual_seed(0)
    r = Runner(cfg, device=dev)
    images, targets = synthetic_batch(2, 512, dev, seed=3, num_gts=24)
    hist = [float(r.train_step(images, targets)[0]) for _ in range(60)]
    print("%-9s loss %.3f -> %.3f (min %.3f), finite %s" % (name, hist[0], hist[-1], min(hist), all(h == h for h in hist)))
    del r
    torch.cuda.empty_cache()
