"""Minimal runner slice: build model / optimizer / scheduler from a config, one train step, the
reference's throughput definition.  Mirrors python/jdet/runner/runner.py: `__init__` L22-70 (registry
builds), `train` L117-155 (losses = model(images, targets); all_loss, losses = parse_losses(losses);
optimizer.step(all_loss); scheduler.step(iter, epoch)), `test_time` L91-115
(FPS = batch * world_size * iters / wall).  val / test / vis / logging back-ends are out of scope.

Data parallelism: one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI); the model
is wrapped in DistributedDataParallel so the gradient all-reduce is bucketed and overlapped with
backward (the reference does a per-parameter all-reduce inside Jittor's Optimizer.pre_step).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

import jdet_amd.models  # noqa: F401  (registers MODELS / BACKBONES / NECKS / HEADS / LOSSES / BOXES)
import jdet_amd.optims  # noqa: F401  (registers OPTIMS / SCHEDULERS)
from jdet_amd.utils.general import parse_losses
from jdet_amd.utils.registry import MODELS, OPTIMS, SCHEDULERS, build_from_cfg


def synthetic_batch(batch, size, device, seed=0, num_gts=64, num_classes=15):
    """Synthetic 1024x1024-style tiles: images N(0,1); per image `num_gts` random OBBs (centre
    U(0,size)^2, w,h = exp(U(ln 16, ln 256)), theta ~ U(-pi/2, pi/2)), labels U{1..15}; target dict keys
    of data/custom.py:L75-88 (SURVEY 8d cfg 2)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rng = np.random.default_rng(seed)
    images = torch.randn((batch, 3, size, size), generator=g).to(device)
    targets = []
    for _ in range(batch):
        c = rng.uniform(0, size, (num_gts, 2))
        wh = np.exp(rng.uniform(np.log(16.0), np.log(min(256.0, size / 2)), (num_gts, 2)))
        th = rng.uniform(-np.pi / 2, np.pi / 2, (num_gts, 1))
        rb = torch.from_numpy(np.concatenate([c, wh, th], 1).astype(np.float32)).to(device)
        # enclosing horizontal boxes (what data/dota.py derives from the polygons) for the two-stage RPNs
        ex = 0.5 * (np.abs(wh[:, :1] * np.cos(th)) + np.abs(wh[:, 1:] * np.sin(th)))
        ey = 0.5 * (np.abs(wh[:, :1] * np.sin(th)) + np.abs(wh[:, 1:] * np.cos(th)))
        hb = torch.from_numpy(np.concatenate([c[:, :1] - ex, c[:, 1:] - ey, c[:, :1] + ex, c[:, 1:] + ey],
                                             1).astype(np.float32)).to(device)
        targets.append(dict(rboxes=rb, hboxes=hb, hboxes_ignore=torch.zeros((0, 4), device=device), labels=torch.from_numpy(rng.integers(1, num_classes + 1, num_gts).astype(np.int32)).to(device),
                            rboxes_ignore=torch.zeros((0, 5), device=device), img_size=(size, size),
                            ori_img_size=(size, size), scale_factor=1.0, pad_shape=(size, size)))
    return images, targets


class Runner:
    def __init__(self, cfg, device=None, ddp=None, channels_last=True, amp_dtype=None, conv_autotune=True):
        self.cfg = cfg
        if conv_autotune:
            # DOTA tiles have one fixed shape: let MIOpen time its solvers once per conv geometry instead of
            # taking the heuristic pick (measured 63.6 -> 58.0 ms per S2ANet step, profiles/r01_miopen_find.txt)
            torch.backends.cudnn.benchmark = True
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world_size > 1 else 0
        self.model = build_from_cfg(cfg["model"], MODELS).to(self.device)
        if channels_last:
            # 4-D conv weights only (ORConv2d keeps a 5-D ARF weight; nn.Module.to(memory_format=) would
            # try to convert it as a 3-D-conv weight and fail)
            for p in self.model.parameters():
                if p.dim() == 4:
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
        self.channels_last = channels_last
        self.amp_dtype = amp_dtype
        opt_cfg = dict(cfg["optimizer"]) if cfg.get("optimizer") else dict(type="SGD", lr=0.0025, momentum=0.9,
                                                                           weight_decay=0.0001,
                                                                           grad_clip=dict(max_norm=35, norm_type=2))
        self.optimizer = build_from_cfg(opt_cfg, OPTIMS, params=list(self.model.parameters()))
        sch_cfg = cfg.get("scheduler")
        self.scheduler = build_from_cfg(dict(sch_cfg), SCHEDULERS, optimizer=self.optimizer) if sch_cfg else None
        use_ddp = self.world_size > 1 if ddp is None else ddp
        self.train_model = self.model
        if use_ddp:
            self.train_model = torch.nn.parallel.DistributedDataParallel(
                self.model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                bucket_cap_mb=int(os.environ.get("JDET_DDP_BUCKET_MB", "64")), gradient_as_bucket_view=True)
        self.iter = 0
        self.epoch = 0

    def train_step(self, images, targets):
        self.train_model.train()
        if self.channels_last and images.dim() == 4:
            images = images.contiguous(memory_format=torch.channels_last)
        if self.amp_dtype is not None:
            with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype):
                losses = self.train_model(images, targets)
        else:
            losses = self.train_model(images, targets)
        all_loss, losses = parse_losses(losses)
        self.optimizer.step(all_loss)
        if self.scheduler is not None:
            self.scheduler.step(self.iter, self.epoch, by_epoch=True)
        self.iter += 1
        return all_loss.detach(), {k: v.detach() for k, v in losses.items()}

    def test_time(self, images, targets, warmup=10, iters=100):
        """the reference's own throughput definition (runner.py:L91-115): FPS = batch*world*iters/wall"""
        for _ in range(warmup):
            self.train_step(images, targets)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            self.train_step(images, targets)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        dt = time.time() - t0
        return images.shape[0] * self.world_size * iters / dt
