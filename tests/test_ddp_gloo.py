"""CPU, world_size 2, gloo: the data-parallel path of jdet_amd.runner.Runner (one process per device,
DDP bucketed gradient all-reduce, identical replicas, image-parallel batches) and bench.py's
barrier + max-over-ranks timing helper.  The detector itself needs a HIP device, so the model here is
a tiny registered conv net -- the plumbing under test (Runner, SGD+clip, StepLR warm-up, DDP, sync) is
the production code."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _register_tiny():
    from jdet_amd.utils.registry import MODELS

    if "TinyDet" in MODELS:
        return

    @MODELS.register_module()
    class TinyDet(nn.Module):
        def __init__(self, width=8):
            super().__init__()
            self.conv = nn.Conv2d(3, width, 3, padding=1)
            self.head = nn.Conv2d(width, 5, 1)

        def forward(self, images, targets):
            y = self.head(torch.relu(self.conv(images)))
            tgt = torch.stack([t["rboxes"].mean(0) for t in targets])  # (N,5)
            return dict(loss_reg=((y.mean((2, 3)) - tgt) ** 2).mean(), loss_aux=[y.abs().mean() * 0.1])


CFG = dict(model=dict(type="TinyDet", width=8),
           optimizer=dict(type="SGD", lr=0.1, momentum=0.9, weight_decay=1e-4, grad_clip=dict(max_norm=35, norm_type=2)),
           scheduler=dict(type="StepLR", warmup="linear", warmup_iters=4, warmup_ratio=1.0 / 3, milestones=[7, 10]))


def _batch(seed, n=2):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn((n, 3, 16, 16), generator=g)
    targets = [dict(rboxes=torch.randn((3, 5), generator=g)) for _ in range(n)]
    return images, targets


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jdet_amd.runner import Runner
        from jdet_amd.utils.general import sync
        _register_tiny()
        torch.manual_seed(7)                      # identical replicas
        r = Runner(CFG, device="cpu", channels_last=False, conv_autotune=False)
        assert r.world_size == world and isinstance(r.train_model, nn.parallel.DistributedDataParallel)
        losses = []
        for it in range(3):
            images, targets = _batch(100 + 10 * it + rank)   # image-parallel: each rank its own tiles
            loss, parts = r.train_step(images, targets)
            losses.append(float(sync(loss)))     # mean over ranks, as general.py:L30-48
        sd = {k: v.clone() for k, v in r.model.state_dict().items()}
        # replicas stay identical after DDP steps
        for k, v in sd.items():
            ref = v.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(ref, v), k
        if rank == 0:
            torch.save(dict(state=sd, losses=losses, lr=r.optimizer.cur_lr()), out)
    finally:
        dist.destroy_process_group()


def test_ddp_matches_single_process_on_the_global_batch(tmp_path):
    port = 29500 + os.getpid() % 2000
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single process, global batch = concatenation of the two ranks' batches: DDP averages gradients over
    # ranks and each rank's loss is a mean over its 2 images -> same update as the mean over all 4
    from jdet_amd.runner import Runner
    _register_tiny()
    torch.manual_seed(7)
    r = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    losses = []
    for it in range(3):
        i0, t0 = _batch(100 + 10 * it + 0)
        i1, t1 = _batch(100 + 10 * it + 1)
        # the aux loss is a mean over all elements and the reg loss a mean over images: both average linearly
        loss, _ = r.train_step(torch.cat([i0, i1]), t0 + t1)
        losses.append(float(loss))
    for k, v in r.model.state_dict().items():
        assert torch.allclose(got["state"][k], v, atol=1e-6), k
    assert all(abs(a - b) < 1e-6 for a, b in zip(got["losses"], losses))
    # StepLR linear warm-up after 3 steps (scheduler.step(iter=2, ...)): lr = base * (1 - (1 - 2/4) * (2/3))
    assert abs(got["lr"] - 0.1 * (1 - (1 - 2 / 4) * (1 - 1 / 3))) < 1e-12


def _timing_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        t = 0.05 if rank == 0 else 0.20           # rank 1 is the slow one
        time.sleep(t)
        tt = torch.tensor([t], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # bench.py: MAX over ranks
        if rank == 0:
            torch.save(float(tt.item()), out)
    finally:
        dist.destroy_process_group()


def test_bench_takes_max_over_ranks(tmp_path):
    port = 31500 + os.getpid() % 2000
    out = str(tmp_path / "t.pt")
    mp.spawn(_timing_worker, args=(2, port, out), nprocs=2, join=True)
    assert abs(torch.load(out) - 0.20) < 1e-9


def test_checkpoint_round_trip_and_reference_style_pickle(tmp_path):
    """Runner.save / load use the reference's checkpoint layout (runner.py:L223-262): a pickle of
    {"meta", "model": name -> numpy array, ...}; a bare name -> array dict (what `jt.save(model.state_dict())` gives)
    loads too; unknown / mis-shaped entries are reported, not fatal."""
    import pickle
    import numpy as np
    from jdet_amd.runner import Runner
    torch.manual_seed(0)
    a = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    a.iter, a.epoch = 37, 2
    path = a.save(str(tmp_path / "ckpt_2.pkl"))
    raw = pickle.load(open(path, "rb"))
    assert set(raw) >= {"meta", "model", "scheduler", "optimizer"}
    assert all(isinstance(v, np.ndarray) for v in raw["model"].values())
    torch.manual_seed(1)
    b = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    missing, unexpected, mismatched = b.load(path)
    assert not missing and not unexpected and not mismatched
    assert (b.iter, b.epoch) == (37, 2)
    for (k, v), (_, w) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(v, w), k
    bare = dict(raw["model"])
    first = next(iter(bare))
    bare["not.a.parameter"] = np.zeros(3, np.float32)
    bare[first] = np.zeros((1, 2, 3), np.float32)
    pickle.dump(bare, open(tmp_path / "bare.pkl", "wb"))
    missing, unexpected, mismatched = b.load(str(tmp_path / "bare.pkl"), model_only=True)
    assert unexpected == ["not.a.parameter"] and mismatched == [first] and not missing


def _fit_worker(rank, world, port, root, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jdet_amd.data import DOTADataset
        from jdet_amd.runner import Runner
        _register_tiny()
        tfm = [dict(type="RotatedResize", min_size=64, max_size=64), dict(type="Pad", size_divisor=32),
               dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)]
        ds = DOTADataset(dataset_dir=root, transforms=tfm, batch_size=2, num_workers=0, shuffle=True)
        seen = []
        orig = ds._read_ann_info
        ds._read_ann_info = lambda idx: (seen.append(ds.img_infos[idx]["filename"]), orig(idx))[1]
        torch.manual_seed(7)
        r = Runner(CFG, device="cpu", channels_last=False, conv_autotune=False)
        loss, _ = r.fit(ds, max_epoch=1)
        assert r.iter == 2 and torch.isfinite(loss)          # 8 images / 2 ranks / batch 2
        for k, v in r.model.state_dict().items():            # replicas identical after training on different shards
            ref = v.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(ref, v), k
        torch.save(sorted(seen), out + ".%d" % rank)
    finally:
        dist.destroy_process_group()


def test_fit_shards_the_dataset_across_ranks(tmp_path):
    """Runner.fit under world_size 2: DistributedSampler gives every rank its own half of the tiles (image-parallel
    weak scaling), DDP keeps the replicas identical"""
    import numpy as np
    from tests.test_data_pipeline import _make_dataset
    root = str(tmp_path / "train")
    _make_dataset(root, [(64, 64)] * 8, np.random.default_rng(2))
    port = 31500 + os.getpid() % 2000
    out = str(tmp_path / "seen")
    mp.spawn(_fit_worker, args=(2, port, root, out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    assert len(a) == len(b) == 4 and not set(a) & set(b) and len(set(a) | set(b)) == 8


def _manual_avg_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jdet_amd.runner import Runner
        _register_tiny()
        torch.manual_seed(7)
        # no DDP wrapper under world_size 2: the path the HIP-graph mode warms up (and falls back) through
        r = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
        assert r.world_size == world and r.train_model is r.model
        for it in range(4):
            images, targets = _batch(100 + 10 * it + rank)
            r.train_step(images, targets)
        for k, v in r.model.state_dict().items():     # gradients were averaged by hand: replicas identical
            ref = v.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(ref, v), k
        for p in r.model.parameters():                 # ... and so are the momentum buffers
            ref = r.optimizer.state[p]["momentum_buffer"].clone()
            dist.broadcast(ref, 0)
            assert torch.equal(ref, r.optimizer.state[p]["momentum_buffer"])
        if rank == 0:
            torch.save({k: v.clone() for k, v in r.model.state_dict().items()}, out)
    finally:
        dist.destroy_process_group()


def test_replicas_without_ddp_wrapper_average_their_gradients(tmp_path):
    """graph mode runs its first iterations (and its fallback) through the eager step WITHOUT a DDP wrapper: the
    gradients must be all-reduced by hand there, or the replicas drift apart for good."""
    port = 33500 + os.getpid() % 2000
    out = str(tmp_path / "m.pt")
    mp.spawn(_manual_avg_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    from jdet_amd.runner import Runner
    _register_tiny()
    torch.manual_seed(7)
    r = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    for it in range(4):
        i0, t0 = _batch(100 + 10 * it + 0)
        i1, t1 = _batch(100 + 10 * it + 1)
        r.train_step(torch.cat([i0, i1]), t0 + t1)
    for k, v in r.model.state_dict().items():
        assert torch.allclose(got[k], v, atol=1e-6), k


def test_resume_continues_the_same_trajectory(tmp_path):
    """save -> load restores counters, scheduler and optimizer state INCLUDING the momentum buffers
    (runner.py:L243-247): three more steps after a resume equal three more steps of the original run"""
    import pickle
    from jdet_amd.runner import Runner
    _register_tiny()
    torch.manual_seed(3)
    a = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    for it in range(3):
        a.train_step(*_batch(it))
    path = a.save(str(tmp_path / "ckpt.pkl"))
    raw = pickle.load(open(path, "rb"))
    assert set(raw["optimizer"]) >= {"lr", "momentum_buffer"} and len(raw["optimizer"]["momentum_buffer"]) == 4
    assert not any(k.endswith("num_batches_tracked") for k in raw["model"])
    torch.manual_seed(99)
    b = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    b.load(path)
    assert b.iter == 3 and abs(b.optimizer.cur_lr() - a.optimizer.cur_lr()) < 1e-12
    for it in range(3, 6):
        la, _ = a.train_step(*_batch(it))
        lb, _ = b.train_step(*_batch(it))
        assert abs(float(la) - float(lb)) < 1e-7
    for (k, v), (_, w) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.allclose(v, w, atol=1e-7), k
