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

from jdet_amd import _lib as L
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
    def __init__(self, cfg, device=None, ddp=None, channels_last=True, amp_dtype=None, conv_autotune=True,
                 graph=None):
        self.cfg = cfg
        # graph=True: the whole step (forward, losses, backward | one flat all-reduce | clip + SGD) replays as
        # HIP graphs; needs a model whose step is fixed-shape and sync-free (the single-stage heads)
        self.use_graph = (os.environ.get("JDET_TRAIN_GRAPH", "0") == "1") if graph is None else bool(graph)
        if conv_autotune:
            # DOTA tiles have one fixed shape: let MIOpen time its solvers once per conv geometry instead of
            # taking the heuristic pick (measured 63.6 -> 58.0 ms per S2ANet step, profiles/r01_miopen_find.txt)
            torch.backends.cudnn.benchmark = True
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type == "cuda":
            if self.device.index is None:   # "cuda" / torch.device("cuda"): the current device
                self.device = torch.device("cuda", torch.cuda.current_device())
            # the C-ABI kernels launch on the CURRENT HIP device: make the runner's device current (one process per GPU)
            torch.cuda.set_device(self.device)
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world_size > 1 else 0
        self.model = build_from_cfg(cfg["model"], MODELS).to(self.device)
        if channels_last:
            # 4-D conv weights only (ORConv2d keeps a 5-D ARF weight; nn.Module.to(memory_format=) would
            # try to convert it as a 3-D-conv weight and fail)
            for p in self.model.parameters():
                # 1x1 kernels are already both layouts; leaving their strides alone keeps them equal to the
                # strides of the gradients MIOpen returns (DDP's bucket views otherwise copy)
                if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
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
        if self.use_graph and self.world_size > 1 and os.environ.get("JDET_TRAIN_GRAPH_MULTI", "1") == "0":
            # Multi-rank HIP-graph steps were refused in rounds 4-5: a replayed step intermittently carried garbage
            # gradients out of memset NODES that do not reliably re-execute (the library's split-K weight-gradient zero
            # fills, the framework's reduction semaphores).  Since round 6 every captured graph is stripped of memset
            # nodes before instantiation (L.harden_graph, csrc/graph_safe.hip) and two-rank graph steps are clean in the
            # diagnosis runs (profiles/r06_graph_notes.md) and strict in tests/test_gpu_ddp_detectors.py;
            # JDET_TRAIN_GRAPH_MULTI=0 keeps the old behaviour (eager DDP steps under world_size > 1).
            if self.rank == 0:
                print("jdet_amd.Runner: JDET_TRAIN_GRAPH_MULTI=0 -- %d ranks -> eager DDP steps" % self.world_size)
            self.use_graph = False
        if self.use_graph and self.device.type == "cuda":
            use_ddp = False    # graph mode all-reduces one flat gradient buffer itself
            if self.world_size > 1:
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    dist.broadcast(t.data, 0)
        if use_ddp:
            self.train_model = torch.nn.parallel.DistributedDataParallel(
                self.model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                bucket_cap_mb=int(os.environ.get("JDET_DDP_BUCKET_MB", "64")), gradient_as_bucket_view=True,
                static_graph=os.environ.get("JDET_DDP_STATIC_GRAPH", "1") == "1",
                # the only buffers are BatchNorm statistics, frozen by `norm_eval` (and the reference has no SyncBN,
                # SURVEY 8e): re-broadcasting them before every forward is pure overhead
                broadcast_buffers=os.environ.get("JDET_DDP_BROADCAST_BUFFERS", "0") == "1")
        self.iter = 0
        self.epoch = 0
        from collections import OrderedDict
        self._graphs = OrderedDict()
        self._graph_cap = int(os.environ.get("JDET_GRAPH_CACHE", "4"))   # captured steps kept (LRU)
        self._graph_misses = 0                                           # consecutive steps that had to capture

    # ------------------------------------------------------------------ HIP-graph step
    def _graph_key(self, images, targets):
        return (tuple(images.shape),) + tuple((k, tuple(v.shape)) for t in targets for k, v in sorted(t.items())
                                              if torch.is_tensor(v))

    def _flat_grads(self):
        """one fp32 buffer holding every gradient; p.grad are views with the parameter's own strides, so
        autograd accumulates in place, the norm / all-reduce see ONE tensor"""
        params = [p for p in self.model.parameters() if p.requires_grad]
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=self.device)
        off = 0
        for p in params:
            n = p.numel()
            if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
                o, i, kh, kw = p.shape
                p.grad = flat[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
            else:
                p.grad = flat[off:off + n].view(p.shape)
            off += n
        return params, flat

    def _capture(self, images, targets):
        st = dict(images=images.clone(), targets=[{k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()}
                                                   for t in targets])
        opt = self.optimizer
        group = opt.param_groups[0]
        st["lr"] = torch.tensor(float(group["lr"]), device=self.device)
        params, flat = self._flat_grads()
        st["flat"] = flat
        bufs = []
        for p in params:
            if opt.state[p].get("momentum_buffer") is None:   # parameter without gradient so far
                opt.state[p]["momentum_buffer"] = torch.zeros_like(p)
            bufs.append(opt.state[p]["momentum_buffer"])
        momentum, wd = group["momentum"], group["weight_decay"]
        clip = opt.grad_clip

        def fwd_bwd():
            L.zero_(flat)          # a plain kernel, not a memset node
            if self.amp_dtype is not None:
                with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
                    losses = self.model(st["images"], st["targets"])
            else:
                losses = self.model(st["images"], st["targets"])
            total, parsed = parse_losses(losses)
            total.backward()
            return total.detach(), {k: v.detach() for k, v in parsed.items()}

        @torch.no_grad()
        def update():
            if clip is not None:   # torch.nn.utils.clip_grad_norm_ on the flat buffer: two kernels
                if float(clip.get("norm_type", 2)) == 2.0:
                    # own two-stage sum (csrc/graph_safe.hip): the framework's vector_norm is a multi-workgroup reduce
                    # whose semaphores are cleared by a memset NODE in the captured graph (scripts/graph_nodes.py)
                    norm = L.norm2(flat)
                else:
                    norm = torch.linalg.vector_norm(flat, float(clip.get("norm_type", 2)))
                flat.mul_(torch.clamp(clip["max_norm"] / (norm + 1e-6), max=1.0))
            grads = [p.grad for p in params]
            if wd != 0:
                grads = torch._foreach_add(grads, params, alpha=wd)
            torch._foreach_mul_(bufs, momentum)
            torch._foreach_add_(bufs, grads)
            torch._foreach_sub_(params, torch._foreach_mul(bufs, st["lr"]))

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):    # capture-stream warm-up (allocator, MIOpen workspaces)
            for _ in range(2):
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        single = self.world_size == 1
        # keep_graph: the captured hipGraph stays editable until its first replay -- every memset node in it (the
        # framework's reduction semaphores, the convolution library's zero fills of its atomically adding solvers) is
        # replaced by a fill-KERNEL node before instantiation (L.harden_graph): memset nodes do not reliably re-execute
        # on replay on this stack, which is what made multi-rank graph steps produce garbage gradients in rounds 4-5
        g1 = L.new_graph()
        with torch.cuda.graph(g1):
            st["out"] = fwd_bwd()
            if single:
                update()
        st["memset_nodes_replaced"] = L.harden_graph(g1)
        st["g1"] = g1
        if not single:
            mode = os.environ.get("JDET_GRAPH_UPDATE", "shared")     # diagnosis switch (scripts/ddp_graph_diag.py)
            if mode == "eager":
                st["g2"] = None
                st["update"] = update
            else:
                g2 = L.new_graph()
                if mode == "own":
                    with torch.cuda.graph(g2):
                        update()
                else:
                    with torch.cuda.graph(g2, pool=g1.pool()):
                        update()
                st["memset_nodes_replaced"] += L.harden_graph(g2)
                st["g2"] = g2
        return st

    def _graph_step(self, images, targets):
        if self.iter < 2:    # momentum buffers, anchor caches and the solver search happen eagerly, once
            return self._eager_step(images, targets)
        key = self._graph_key(images, targets)
        st = self._graphs.get(key)
        if st is None:
            # The key holds every target shape (the per-image gt counts): a real dataset gives a new key almost every
            # step, and each capture costs two warm-up passes, a private memory pool and a flat gradient buffer.
            # Graph mode is for fixed-shape batches: after a few consecutive misses it is switched off; the cache is
            # bounded (least recently used capture dropped).
            self._graph_misses += 1
            if self._graph_misses > max(2, self._graph_cap):
                # (ranks decide on their own: batch signatures differ per rank.  That is safe because an eager step's
                # all-reduce -- _allreduce_grads -- uses the flat layout of the graph step, element for element)
                if self.rank == 0:
                    print("jdet_amd.Runner: %d consecutive steps with a new batch signature -- HIP-graph mode is for "
                          "fixed-shape batches; continuing with eager steps" % self._graph_misses)
                self._leave_graph_mode()
                return self._eager_step(images, targets)
            self.model.train()
            while len(self._graphs) >= self._graph_cap:
                self._graphs.popitem(last=False)
            st = self._graphs[key] = self._capture(images, targets)
        else:
            self._graph_misses = 0
            self._graphs.move_to_end(key)
            st["images"].copy_(images, non_blocking=True)
            for dst, src in zip(st["targets"], targets):
                for k, v in src.items():
                    if torch.is_tensor(v):
                        dst[k].copy_(v, non_blocking=True)
        st["lr"].fill_(float(self.optimizer.param_groups[0]["lr"]))
        st["g1"].replay()
        if self.world_size > 1:
            self._allreduce_flat(st["flat"])
            if st["g2"] is not None:
                st["g2"].replay()
            else:
                st["update"]()
        if self.scheduler is not None:
            self.scheduler.step(self.iter, self.epoch, by_epoch=True)
        self.iter += 1
        return st["out"]

    def train_step(self, images, targets):
        if self.use_graph and self.device.type == "cuda":
            if self.channels_last and images.dim() == 4:
                images = images.contiguous(memory_format=torch.channels_last)
            return self._graph_step(images, targets)
        return self._eager_step(images, targets)

    def _eager_step(self, images, targets):
        self.train_model.train()
        if self.channels_last and images.dim() == 4:
            images = images.contiguous(memory_format=torch.channels_last)
        if self.amp_dtype is not None:
            with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype):
                losses = self.train_model(images, targets)
        else:
            losses = self.train_model(images, targets)
        all_loss, losses = parse_losses(losses)
        if self.world_size > 1 and self.train_model is self.model:
            # no DDP wrapper (graph mode warms up / falls back through here): the replicas stay identical only if the
            # gradients are averaged before the update, exactly as the graph step does with its flat buffer
            self.optimizer.zero_grad(set_to_none=True)
            all_loss.backward()
            self._allreduce_grads()
            self.optimizer.step(None)
        else:
            self.optimizer.step(all_loss)
        if self.scheduler is not None:
            self.scheduler.step(self.iter, self.epoch, by_epoch=True)
        self.iter += 1
        return all_loss.detach(), {k: v.detach() for k, v in losses.items()}

    def _allreduce_flat(self, flat):
        """mean over ranks, in place.  RCCL orders the collective on the current stream.  The gloo backend (CPU tests, the
        two-ranks-on-one-GPU test) would stage a device tensor through pinned host memory on streams of its own; that
        path gave intermittently different results on the two ranks here (tests/test_gpu_ddp_detectors.py, graph mode,
        ~1 run in 4), so for gloo the staging is done here, synchronously"""
        if flat.is_cuda and dist.get_backend() == "gloo":
            host = flat.cpu()
            dist.all_reduce(host)
            flat.copy_(host)
        else:
            dist.all_reduce(flat)
        flat.div_(self.world_size)

    @staticmethod
    def _flat_order(p, t):
        """`t` (shaped like parameter p) in the element order `_flat_grads` gives p's segment of the flat buffer:
        channels-last 4-D parameters are laid out (o, kh, kw, i), everything else in logical order"""
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            return t.permute(0, 2, 3, 1)
        return t

    def _allreduce_grads(self):
        """Gradient all-reduce of an eager step without DDP.  Same buffer as a graph step's, element for element:
        every trainable parameter in model order (zeros where a rank has no gradient), so a rank that has left
        HIP-graph mode and one that still replays its capture take part in the same collective."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        if not params:
            return
        flat = torch.cat([(self._flat_order(p, p.grad).reshape(-1) if p.grad is not None
                           else torch.zeros(p.numel(), dtype=torch.float32, device=p.device)) for p in params])
        self._allreduce_flat(flat)
        off = 0
        for p in params:
            n = p.numel()
            seg = flat[off:off + n]
            off += n
            if p.grad is None:       # the other ranks' contribution
                p.grad = torch.zeros_like(p)
            view = self._flat_order(p, p.grad)
            view.copy_(seg.view(view.shape))

    def _leave_graph_mode(self):
        """drop the captures and continue eagerly (p.grad no longer aliases a flat capture buffer)"""
        self.use_graph = False
        self._graphs.clear()
        for p in self.model.parameters():
            p.grad = None

    # ------------------------------------------------------------------ training on a dataset
    def fit(self, dataset, max_epoch=1, max_iter=None, log_interval=0):
        """runner.py:L117-155 over a `jdet_amd.data` dataset: one process per GPU reads its own shard
        (DistributedSampler), batches are staged to the device one step ahead on a side stream (DeviceFeeder),
        `scheduler.step(iter, epoch, by_epoch=True)` after every iteration.  Returns the last (loss, parts)."""
        from torch.utils.data.distributed import DistributedSampler
        from jdet_amd.data import DeviceFeeder
        if self.use_graph:
            # batches of a dataset differ in their gt counts: every step would re-capture (see _graph_step)
            if self.rank == 0:
                print("jdet_amd.Runner.fit: HIP-graph mode needs fixed-shape batches; using eager steps")
            self._leave_graph_mode()
        sampler = DistributedSampler(dataset, shuffle=dataset.shuffle) if self.world_size > 1 else None
        feeder = DeviceFeeder(dataset.loader(sampler=sampler), self.device)
        last = None
        for epoch in range(self.epoch, max_epoch):
            self.epoch = epoch
            if sampler is not None:
                sampler.set_epoch(epoch)
            for images, targets in feeder:
                last = self.train_step(images, targets)
                if log_interval and self.rank == 0 and self.iter % log_interval == 0:
                    print("epoch %d iter %d lr %.6f loss %.4f" % (epoch, self.iter, self.optimizer.cur_lr(),
                                                                  float(last[0])))
                if max_iter is not None and self.iter >= max_iter:
                    return last
        self.epoch = max_epoch
        return last

    # ------------------------------------------------------------------ checkpoints
    # runner.py:L223-262.  `jt.save` of a `.pkl` is a pickle of plain python containers with every Var turned into
    # a numpy array, so reference checkpoints load without Jittor and ours load there: {"meta", "model" (parameter
    # name -> array, same names as the reference's modules), "scheduler", "optimizer"}.
    def save(self, path):
        import pickle
        if self.rank != 0:
            return None
        data = {
            "meta": {"jdet_version": "jdet_amd", "epoch": self.epoch, "iter": self.iter,
                     "save_time": time.strftime("%Y-%m-%d %H:%M:%S")},
            # torch-only bookkeeping buffers are not parameters of the reference's modules
            "model": {k: v.detach().cpu().numpy() for k, v in self.model.state_dict().items()
                      if not k.endswith("num_batches_tracked")},
            "scheduler": self.scheduler.parameters() if self.scheduler is not None and
            hasattr(self.scheduler, "parameters") else {},
            "optimizer": self._optimizer_state(),
        }
        data["meta"].update({"max_iter": self.cfg.get("max_iter"), "max_epoch": self.cfg.get("max_epoch")}
                            if hasattr(self.cfg, "get") else {})
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "wb") as f:
            pickle.dump(data, f)
        return path

    def load(self, path, model_only=False):
        """returns (missing, unexpected, mismatched) parameter names; like `load_parameters` in the reference, names
        that do not line up are reported, not fatal"""
        import pickle
        with open(path, "rb") as f:
            data = pickle.load(f)
        if not model_only and isinstance(data, dict):
            # runner.py:L243-247: resume = meta counters + scheduler + optimizer state
            meta = data.get("meta", {})
            self.epoch = meta.get("epoch", self.epoch)
            self.iter = meta.get("iter", self.iter)
            if self.scheduler is not None and data.get("scheduler") and hasattr(self.scheduler, "load_parameters"):
                self.scheduler.load_parameters(data["scheduler"])
            if data.get("optimizer"):
                self._load_optimizer_state(data["optimizer"])
        if isinstance(data, dict) and "model" in data:
            params = data["model"]
        elif isinstance(data, dict) and "state_dict" in data:
            params = data["state_dict"]
        else:
            params = data
        own = self.model.state_dict()
        missing = [k for k in own if k not in params and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in params if k not in own]
        mismatched, good = [], {}
        for k, v in params.items():
            if k not in own:
                continue
            t = torch.as_tensor(np.asarray(v))
            if tuple(t.shape) != tuple(own[k].shape):
                mismatched.append(k)
                continue
            good[k] = t.to(own[k].dtype)
        # through load_state_dict: modules that keep a parameter in another element order than the reference
        # (RoIFeatureLinear) convert in their load hooks; copies keep each parameter's own memory format
        self.model.load_state_dict(good, strict=False)
        return missing, unexpected, mismatched

    def _optimizer_state(self):
        """lr + one momentum buffer per trainable parameter NAME (numpy), so that a resumed run continues the
        same trajectory (optimizer.py:L8-36 keeps the same state in Jittor's optimizer)"""
        names = {id(p): n for n, p in self.model.named_parameters()}
        mom = {}
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                buf = self.optimizer.state.get(p, {}).get("momentum_buffer")
                if buf is not None and id(p) in names:
                    mom[names[id(p)]] = buf.detach().cpu().numpy()
        return {"lr": self.optimizer.cur_lr(), "momentum_buffer": mom}

    def _load_optimizer_state(self, st):
        if not isinstance(st, dict):
            return
        if "lr" in st:
            for g in self.optimizer.param_groups:
                g["lr"] = float(st["lr"])
            if self.scheduler is not None:
                # one saved rate cannot restore per-group rates (base_lr_pg multipliers): recompute every group's
                # rate from the schedule at the restored position, as the next step would
                self.scheduler.step(max(self.iter - 1, 0), self.epoch, by_epoch=True)
        params = dict(self.model.named_parameters())
        with torch.no_grad():
            for n, arr in (st.get("momentum_buffer") or {}).items():
                p = params.get(n)
                if p is None or tuple(np.shape(arr)) != tuple(p.shape):
                    continue
                buf = torch.zeros_like(p)
                buf.copy_(torch.as_tensor(np.asarray(arr)).to(p.dtype))
                self.optimizer.state[p]["momentum_buffer"] = buf

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
