#!/usr/bin/env python3
"""bench.py -- driver contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.

Default workload `s2anet_train` = BASELINE.json's headline: S2ANet-R50-FPN train step on synthetic 1024x1024
tiles, 2 images per GPU, 64 random OBB gts per image, random-init weights of the reference architecture
(configs/s2anet/s2anet_r50_fpn_1x_dota.py), fp32 like the reference, SGD + clip + StepLR, DDP/RCCL gradient
all-reduce when launched on N > 1 ranks (weak scaling: per-GPU batch fixed).  `value` = img/s over all ranks.
The same line carries
  * `roofline`: the path's HBM-bound hand-written kernel, rotated RoIAlign forward at the north-star point
    (1x256x256x256 fp32 map = stride-4 level of a 1024x1024 tile, 2000 RoIs, 7x7, sampling 2), measured live in
    this process with HIP events on the launch stream.  ALGORITHMIC bytes per launch:
    4*N*C*H*W + 4*R*C*PH*PW + 24*R = 67.11 + 100.35 + 0.05 MB = 167.5 MB;
  * `cpu_baseline`: the CPU oracle (kind "port") on that same RoIAlign workload on all host cores.

Other workloads (`--workload`): roi_align_rotated (the roofline leg as its own line), roi_align_rotated_bwd,
box_iou_rotated, nms_rotated, retinanet_infer (BASELINE configs[1]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None)
    p.add_argument("--warmup", type=int, default=None)
    p.add_argument("--workload", default="s2anet_train")
    p.add_argument("--batch", type=int, default=None,
                   help="images per GPU (train workloads); default: the BASELINE.json config's -- 2, roitrans_train (cfg 4) 4")
    p.add_argument("--size", type=int, default=1024, help="tile size (s2anet_train)")
    p.add_argument("--amp", default="none", choices=["none", "bf16", "fp16"])
    p.add_argument("--rois", type=int, default=2000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-live-traffic", action="store_true",
                   help="roofline.traffic from profiles/hbm_traffic.json instead of two rocprofv3 --pmc passes run now")
    p.add_argument("--no-secondary", action="store_true",
                   help="leave out the `secondary` object (ms/step of the other BASELINE.json model configs)")
    a = p.parse_args()
    if a.batch is None:
        a.batch = 4 if a.workload == "roitrans_train" else 2      # configs[4]: batch 32 over 8 GPUs
    model_level = a.workload in ("s2anet_train", "retinanet_infer")
    if a.steps is None:
        a.steps = 20 if model_level else 200
    if a.warmup is None:
        a.warmup = 5 if model_level else 20
    return a


def make_inputs(workload, R, seed, dev):
    from tests import inputs as I
    rng = np.random.default_rng(seed)
    d = {}
    if workload.startswith("roi_align") or workload == "riroi_align":
        g = torch.Generator(device="cpu").manual_seed(seed)
        feat = torch.randn((1, 256, 256, 256), generator=g)
        d["feat_np"] = None
        d["feat"] = feat.to(dev).contiguous(memory_format=torch.channels_last)
        rois = I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))
        d["rois_np"] = rois
        d["rois"] = torch.from_numpy(rois).to(dev)
        d["grad"] = torch.randn((R, 256, 7, 7), device=dev) if ("bwd" in workload or workload.endswith("pair")) else None
        d["feat_cpu"] = feat
    elif workload == "box_iou_rotated":
        b1, b2 = I.random_obbs(rng, 64, wh=(16.0, 256.0)), I.random_obbs(rng, 21824)
        d["b1_np"], d["b2_np"] = b1, b2
        d["b1"], d["b2"] = torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev)
    elif workload == "nms_rotated":
        n = R
        dets = np.concatenate([I.random_obbs(rng, n // 2), I.clustered_obbs(rng, n - n // 2, 64, 1024.0)], 0)
        scores = (rng.uniform(0, 1, n) + np.arange(n) * 1e-7).astype(np.float32)
        d["dets_np"], d["scores_np"] = dets, scores
        d["dets"], d["scores"] = torch.from_numpy(dets).to(dev), torch.from_numpy(scores).to(dev)
        d["order"] = torch.argsort(d["scores"], descending=True, stable=True)
    else:
        raise SystemExit("unknown workload " + workload)
    return d


def make_step(workload, d):
    """returns (step_fn, units_per_step, unit_name, algorithmic_bytes_per_launch, kernel_name, dtype)"""
    from jdet_amd import _lib as L
    lib = L.lib()
    if workload == "roi_align_rotated":
        feat, rois = d["feat"], d["rois"]
        R = rois.shape[0]
        nbytes = 4 * 256 * 256 * 256 + 4 * R * 256 * 49 + 24 * R
        fp, rp = feat.data_ptr(), rois.data_ptr()
        path = os.environ.get("JDET_ROI_FWD_PATH", "roi_cl")
        if path == "pool":
            # EXPERIMENTAL (libjdet_experimental.so, not a product path): schedule + plan + register-cached pool kernel
            from jdet_amd import _experimental as X
            xl = X.lib()
            out = torch.empty((R, 256, 7, 7), device=feat.device, memory_format=torch.channels_last)
            op = out.data_ptr()
            wsb = xl.jdet_roi_align_forward_pool_workspace(R)
            ws = torch.empty((wsb,), dtype=torch.uint8, device=feat.device)
            wp = ws.data_ptr()

            def step():
                L.check(xl.jdet_roi_align_forward_pool(0, fp, 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, op, wp, wsb,
                                                       L.stream_ptr(feat)), "fwd_pool")
            d["out"] = out
            return (step, nbytes / 1e9, "GB", nbytes,
                    "EXPERIMENTAL roi_order_kernel + roi_plan_kernel<ROTATED> + roi_pool_kernel (channels-last output)",
                    "f32")
        if path in ("sliced", "line", "staged"):
            # EXPERIMENTAL (libjdet_experimental.so, not product paths): the channel-sliced plan + pool kernels / the
            # line-deduplicating kernel of round 4 (profiles/r04_roi_fwd_notes.md)
            from jdet_amd import _experimental as X
            xl = X.lib()
            mode = {"sliced": 2, "line": 3, "staged": 4}[path]   # staged: round 6, csrc/roi_align_stage.h (LDS-DMA)
            out = torch.empty((R, 256, 7, 7), device=feat.device, memory_format=torch.channels_last)
            op = out.data_ptr()
            wsb = xl.jdet_roi_align_forward_cl_mode_workspace(mode, R, 7, 7)
            ws = torch.empty((wsb,), dtype=torch.uint8, device=feat.device)
            wp = ws.data_ptr()
            obuf = torch.empty((2, R), dtype=torch.int32, device=feat.device)
            o0, o1 = obuf[0].data_ptr(), obuf[1].data_ptr()
            if path == "sliced" and os.environ.get("JDET_ROI_SLICED_PLANAR", "0") == "1":
                # EXPERIMENT (L2 channel spread): the map as [slice][pixel][32 channels]; same values, other addresses
                planar = feat.permute(0, 2, 3, 1).reshape(256 * 256, 8, 32).permute(1, 0, 2).contiguous()
                d["planar"] = planar
                fp = planar.data_ptr()

            def step():
                st = L.stream_ptr(feat)
                if mode >= 3:
                    L.check(lib.jdet_roi_spatial_order(rp, R, 6, 0.25, 1, 256, 256, o0, o1, st), "order")
                L.check(xl.jdet_roi_align_forward_cl_mode(mode, 0, fp, 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, 1,
                                                          o0 if mode >= 3 else None, op, wp, wsb, st), "fwd_cl_mode")
            d["out"] = out
            return (step, nbytes / 1e9, "GB", nbytes,
                    "EXPERIMENTAL roi_sort_plan_kernel<ROTATED> + roi_pool_sliced_kernel (channels-last out)" if mode == 2
                    else "EXPERIMENTAL roi_order_kernel + roi_align_fwd_line_kernel<ROTATED> (channels-last out)" if mode == 3
                    else "EXPERIMENTAL roi_order_kernel + roi_align_fwd_staged_kernel<ROTATED> (LDS-DMA, channels-last out)", "f32")
        if path == "roi_cl":   # default product path: jdet_roi_align_forward_cl = XCD-aware schedule + RoI-stationary kernel
            out = torch.empty((R, 256, 7, 7), device=feat.device, memory_format=torch.channels_last)
            op = out.data_ptr()
            wsb = lib.jdet_roi_align_forward_cl_workspace(R, 7, 7)
            ws = torch.empty((wsb,), dtype=torch.uint8, device=feat.device)
            wp = ws.data_ptr()

            def step():
                # the schedule is recomputed every step: RoIs arrive in arbitrary order
                L.check(lib.jdet_roi_align_forward_cl(0, fp, 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, 1, op, wp, wsb,
                                                      L.stream_ptr(feat)), "fwd_cl")
            d["out"] = out
            return (step, nbytes / 1e9, "GB", nbytes,
                    "roi_order_kernel + roi_align_fwd_merged_kernel<ROTATED,4 waves,channels-last out,rolling window>", "f32")
        # path "roi": the RoI-stationary kernels with the reference's (R,C,7,7)-contiguous result
        cl = False
        out = torch.empty((R, 256, 7, 7), device=feat.device)
        obuf = torch.empty((2, R), dtype=torch.int32, device=feat.device)
        op = out.data_ptr()
        o0, o1 = obuf[0].data_ptr(), obuf[1].data_ptr()
        use_order = os.environ.get("JDET_BENCH_NO_ORDER", "0") != "1"

        def step():
            st = L.stream_ptr(feat)
            if use_order:
                L.check(lib.jdet_roi_spatial_order(rp, R, 6, 0.25, 1, 256, 256, o0, o1, st), "order")
            L.check(lib.jdet_roi_align_forward(0, fp, 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, 1,
                                               o0 if use_order else None, op, st), "fwd")
        d["out"] = out
        return (step, nbytes / 1e9, "GB", nbytes, "roi_order_kernel + roi_align_fwd_merged_kernel<ROTATED,4 waves,(R,C,7,7) out>",
                "f32")
    if workload == "riroi_align":
        # RiRoIAlign forward (riroi_align.py:L70-173) at the north-star shape: 256 planes = 32 channels x 8 orientations
        feat, rois = d["feat"], d["rois"]
        R = rois.shape[0]
        nbytes = 4 * 256 * 256 * 256 + 4 * R * 256 * 49 + 24 * R
        out = torch.empty((R, 256, 7, 7), device=feat.device, memory_format=torch.channels_last)
        wsb = lib.jdet_roi_align_forward_cl_workspace(R, 7, 7)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=feat.device)
        fp, rp, op, wp = feat.data_ptr(), rois.data_ptr(), out.data_ptr(), ws.data_ptr()

        def step():
            L.check(lib.jdet_roi_align_forward_cl(2, fp, 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, 8, op, wp, wsb,
                                                  L.stream_ptr(feat)), "riroi_fwd_cl")
        d["out"] = out
        return (step, nbytes / 1e9, "GB", nbytes,
                "roi_order_kernel + roi_align_fwd_merged_kernel<ROTATED,4 waves,channels-last out,nO=8,rolling window>", "f32")
    if workload == "roi_align_rotated_bwd":
        feat, rois, grad = d["feat"], d["rois"], d["grad"]
        R = rois.shape[0]
        gin = torch.empty_like(feat)
        cl = os.environ.get("JDET_BENCH_BWD_LAYOUT", "cl") == "cl"   # grad_out channels-last (what the tile forward pairs with)
        if cl:
            grad = grad.contiguous(memory_format=torch.channels_last)
        gp, rp, ip = grad.data_ptr(), rois.data_ptr(), gin.data_ptr()
        use_ws = os.environ.get("JDET_BENCH_BWD_ATOMIC", "0") != "1"
        wsb = lib.jdet_roi_align_backward_workspace(0, R, 1, 256, 256, 256, 7, 7, 2) if use_ws else 0
        # kept workspace, zero-filled once: the call hands its counters back zeroed (what the autograd path does)
        ws = torch.zeros((max(wsb, 8),), dtype=torch.uint8, device=feat.device)
        wp = ws.data_ptr() if wsb else None

        def step():
            if cl and wsb:
                L.check(lib.jdet_roi_align_backward_cl(0, gp, rp, R, 1, 256, 256, 256, 7, 7, 0.25, 2, 1, ip, wp, wsb, 1,
                                                       L.stream_ptr(feat)), "bwd_cl")
            else:
                L.check(lib.jdet_roi_align_backward(0, gp, rp, R, 1, 256, 256, 256, 7, 7, 0.25, 2, 1, None, ip,
                                                    wp, wsb, L.stream_ptr(feat)), "bwd")
        nbytes = 4 * 256 * 256 * 256 + 4 * R * 256 * 49 + 24 * R
        return step, nbytes / 1e9, "GB", nbytes, "roi_align backward (taps+scan+fill%s+gather | atomic)" % ("" if cl else "+zero+transpose"), "f32"
    if workload in ("roi_align_rotated_bwd_planned", "roi_align_rotated_pair"):
        # round 6: the backward's plan (jdet_roi_align_backward_plan: the inversion of the scatter) is built ONCE per RoI
        # set -- at the forward in training -- and the backward is the gather alone (jdet_roi_align_backward_cl_planned).
        #   ..._bwd_planned : a step = the gather from a kept plan (the plan's build is outside the timed region)
        #   ..._pair        : a step = what one training step does with one RoI set: schedule + forward kernel, the plan
        #                     built beside it on a second stream, then the gather
        feat, rois = d["feat"], d["rois"]
        R = rois.shape[0]
        grad = d["grad"].contiguous(memory_format=torch.channels_last)
        gin = torch.empty_like(feat)
        out = torch.empty((R, 256, 7, 7), device=feat.device, memory_format=torch.channels_last)
        pb = lib.jdet_roi_align_backward_plan_bytes(0, R, 1, 256, 256, 7, 7, 2)
        plan = torch.empty((pb,), dtype=torch.uint8, device=feat.device)
        wsb = lib.jdet_roi_align_forward_cl_workspace(R, 7, 7)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=feat.device)
        fp, rp, gp, ip, op, pp, wp = (feat.data_ptr(), rois.data_ptr(), grad.data_ptr(), gin.data_ptr(), out.data_ptr(),
                                      plan.data_ptr(), ws.data_ptr())
        nbytes = 4 * 256 * 256 * 256 + 4 * R * 256 * 49 + 24 * R
        L.check(lib.jdet_roi_align_backward_plan(0, rp, R, 1, 256, 256, 7, 7, 0.25, 2, pp, pb, L.stream_ptr(feat)), "plan")
        if workload == "roi_align_rotated_bwd_planned":
            def step():
                L.check(lib.jdet_roi_align_backward_cl_planned(0, gp, R, 1, 256, 256, 256, 7, 7, 2, ip, pp, pb,
                                                               L.stream_ptr(feat)), "bwd_planned")
            return step, nbytes / 1e9, "GB", nbytes, "csr_gather_patch_kernel<8, keep plan> (plan kept from the forward)", "f32"
        side = torch.cuda.Stream(feat.device)
        ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()

        def step():
            cur = torch.cuda.current_stream(feat.device)
            ev_fork.record(cur)
            side.wait_event(ev_fork)
            L.check(lib.jdet_roi_align_backward_plan(0, rp, R, 1, 256, 256, 7, 7, 0.25, 2, pp, pb, side.cuda_stream), "plan")
            ev_join.record(side)
            L.check(lib.jdet_roi_align_forward_cl(0, fp, 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, 1, op, wp, wsb,
                                                  cur.cuda_stream), "fwd_cl")
            cur.wait_event(ev_join)
            L.check(lib.jdet_roi_align_backward_cl_planned(0, gp, R, 1, 256, 256, 256, 7, 7, 2, ip, pp, pb,
                                                           cur.cuda_stream), "bwd_planned")
        d["out"] = out
        return (step, 2 * nbytes / 1e9, "GB", 2 * nbytes,
                "forward (roi_order + roi_align_fwd_merged) || plan (bwd_patch_taps + scan + fill), then csr_gather_patch", "f32")
    if workload == "box_iou_rotated":
        from jdet_amd.ops import box_iou_rotated
        b1, b2 = d["b1"], d["b2"]

        def step():
            d["out"] = box_iou_rotated(b1, b2)
        n = b1.shape[0] * b2.shape[0]
        return step, n / 1e6, "Mpair", 20 * (b1.shape[0] + b2.shape[0]) + 4 * n, "box_iou_kernel", "f32"
    if workload == "nms_rotated":
        from jdet_amd.ops.nms_rotated import nms_rotated_keep_mask
        dets, order = d["dets"], d["order"]

        def step():
            d["out"] = nms_rotated_keep_mask(dets, order, 0.1)
        n = dets.shape[0]
        return step, n * (n - 1) / 2 / 1e6, "Mpair", 20 * n + 8 * n * ((n + 63) // 64) + n, "nms_mask_kernel", "f32"
    raise SystemExit("unknown workload")




from jdet_amd.config.named import ORCNN_CFG, RETINANET_CFG, S2ANET_CFG, roitrans_train_cfg  # noqa: E402


TRAIN_WORKLOADS = {
    # name: (config, model title, images per GPU in BASELINE.json, config-file note)
    "s2anet_train": (lambda: S2ANET_CFG, "S2ANet-R50-FPN", "configs/s2anet/s2anet_r50_fpn_1x_dota.py"),
    "orcnn_train": (lambda: ORCNN_CFG, "Oriented-RCNN R50-FPN", "configs/oriented_rcnn_r50_fpn_1x_dota_with_flip.py"),
    "roitrans_train": (lambda: roitrans_train_cfg("Resnet101"), "RoI-Transformer R101-FPN",
                       "configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py with Resnet101 (SURVEY 8d cfg 4)"),
    "roitrans_r50_train": (lambda: roitrans_train_cfg("Resnet50"), "RoI-Transformer R50-FPN",
                           "configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py"),
}


def make_train(a, rank, dev):
    """One train step of a named config (SURVEY 8d cfg 2/3/4): `batch` synthetic tiles per GPU, 64 random OBB gts
    each, random-init weights of the reference architecture, SGD + clip + StepLR, DDP (RCCL) when world > 1."""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.runner import Runner, synthetic_batch
    torch.manual_seed(1234)  # identical replicas
    amp = {"none": None, "bf16": torch.bfloat16, "fp16": torch.float16}[a.amp]
    runner = Runner(TRAIN_WORKLOADS[a.workload][0](), device=dev, amp_dtype=amp,
                    conv_autotune=os.environ.get("JDET_CUDNN_BENCHMARK", "1") == "1",
                    ddp=True if os.environ.get("JDET_BENCH_FORCE_DIST", "0") == "1" else None)
    images, targets = synthetic_batch(a.batch, a.size, dev, seed=2 + rank)
    images = images.contiguous(memory_format=torch.channels_last)

    def step():
        runner.train_step(images, targets)
    return step, runner


def make_retinanet_infer(a, rank, dev):
    """RetinaNet-OBB R50-FPN inference (BASELINE configs[1]): 1 x 3 x 1024 x 1024 synthetic tile, random-init
    weights.  Random weights put every score at ~0.01 < score_thr, so (SURVEY 8d cfg 1) the classification logits
    are shifted to N(-6, 1.5)-like values by a fixed additive tensor: ~2-3 k candidates reach rotated NMS."""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    torch.manual_seed(1)
    model = build_from_cfg(RETINANET_CFG["model"], MODELS).to(dev).eval()
    for p in model.parameters():
        if p.dim() == 4:
            p.data = p.data.contiguous(memory_format=torch.channels_last)
    images, targets = synthetic_batch(a.batch if a.workload != "retinanet_infer" else 1, a.size, dev, seed=1 + rank)
    images = images.contiguous(memory_format=torch.channels_last)
    head = model.bbox_head
    g = torch.Generator(device="cpu").manual_seed(7)
    noise = {}
    orig = head.get_bboxes

    def get_bboxes(cls_scores, bbox_preds, img_metas, rescale=True):
        shifted = []
        for lvl, cls in enumerate(cls_scores):
            if lvl not in noise:
                noise[lvl] = (torch.randn(cls.shape, generator=g) * 1.5 - 1.4).to(dev)   # bias init is -4.6
            shifted.append(cls + noise[lvl])
        return orig(shifted, bbox_preds, img_metas, rescale)
    head.get_bboxes = get_bboxes
    amp = {"none": None, "bf16": torch.bfloat16, "fp16": torch.float16}[a.amp]

    def step():
        with torch.no_grad():
            if amp is not None:
                with torch.autocast(device_type="cuda", dtype=amp):
                    return model(images, targets)
            return model(images, targets)
    return step, model


def cpu_baseline(workload, d, R):
    """CPU baseline on a bounded sample of the same workload.  RoIAlign has no CPU implementation in the reference
    (CUDA kernels only): kind "port" = the oracle restatement, all host cores (OpenMP over RoIs).  Rotated IoU / NMS:
    kind "reference" = the reference's own CPU source compiled for the host (oracle/_ref), one thread as there."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    O.set_threads(cores)
    O.lib()
    if workload.startswith("roi_align"):
        feat = d["feat_cpu"].numpy()
        # size the sample from a 16-RoI probe so that it costs ~10 s (many-core hosts finish the
        # whole workload in well under a second: then repeat it and keep the mean)
        t0 = time.perf_counter()
        O.roi_align_forward(O.V_ROT, feat, d["rois_np"][:16], (7, 7), 0.25, 2)
        per = (time.perf_counter() - t0) / 16
        rs = int(min(R, max(32, 10.0 / max(per, 1e-6))))
        bwd = workload.endswith("bwd")
        g = np.ones((rs, 256, 7, 7), np.float32) if bwd else None
        cores_used = 1 if bwd else cores  # backward accumulates serially (as the oracle defines it)
        reps, t = 0, 0.0
        while reps < 1 or (t < 5.0 and reps < 50):
            t0 = time.perf_counter()
            if bwd:
                O.roi_align_backward(O.V_ROT, g, d["rois_np"][:rs], feat.shape, 0.25, 2)
            else:
                O.roi_align_forward(O.V_ROT, feat, d["rois_np"][:rs], (7, 7), 0.25, 2)
            t += time.perf_counter() - t0
            reps += 1
        t /= reps
        full = t * R / rs
        nbytes = 4 * 256 * 256 * 256 + 4 * R * 256 * 49 + 24 * R
        out = {"value": nbytes / 1e9 / full, "unit": "GB/s", "cores": cores_used, "kind": "port",
               "sample": "first %d of %d RoIs of the same map, mean of %d runs (%.3f s each), extrapolated "
                         "linearly in RoIs; algorithmic bytes of the full workload / extrapolated time"
                         % (rs, R, reps, t)}
        return out
    if workload == "box_iou_rotated":
        b1, b2 = d["b1_np"][:16], d["b2_np"]
        if O.have_ref():   # the reference's true CPU source (box_iou_rotated.py:L312-326, L487-500), one thread
            t0 = time.perf_counter()
            O.ref_box_iou_rotated(b1, b2)
            t = time.perf_counter() - t0
            return {"value": b1.shape[0] * b2.shape[0] / 1e6 / t, "unit": "Mpair/s", "cores": 1, "kind": "reference",
                    "sample": "16 of 64 gt rows x 21824 anchors through the reference's CPU text (%.2f s)" % t}
        t0 = time.perf_counter()
        O.box_iou_rotated(b1, b2)
        t = time.perf_counter() - t0
        return {"value": b1.shape[0] * b2.shape[0] / 1e6 / t, "unit": "Mpair/s", "cores": min(cores, 16),
                "kind": "port", "sample": "16 of 64 gt rows x 21824 anchors (%.1f s)" % t}
    if workload == "nms_rotated":
        dets, scores = d["dets_np"], d["scores_np"]
        order = np.argsort(-scores, kind="stable").astype(np.int32)
        kind = "reference" if O.have_ref() else "port"   # reference: nms_rotated.py:L314-328, L414-449 CPU text
        t0 = time.perf_counter()
        (O.ref_nms_rotated_keep if kind == "reference" else O.nms_rotated_keep)(dets, order, 0.1)
        t = time.perf_counter() - t0
        n = dets.shape[0]
        return {"value": n * (n - 1) / 2 / 1e6 / t, "unit": "Mpair/s", "cores": 1, "kind": kind,
                "sample": "full n=%d greedy NMS, single thread as in the reference (%.1f s); upper-triangle "
                          "pairs / time (the greedy loop skips suppressed rows)" % (n, t)}
    return None


def timed(step, steps, warmup, dist, dev):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; returns
    (wall seconds = max over ranks, device ms per step from HIP events on the launch stream)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / steps  # HIP events on torch's current stream = the launch stream
    if dist is not None:
        tt = torch.tensor([t], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    return t, dev_ms


_COPY_PEAK = {}


def measured_copy_peak(dev, gib=1, reps=20):
    """HBM rate of THIS device, measured on the lease: a device-to-device copy of `gib` GiB (far beyond the 256 MB
    Infinity Cache), `reps` timed repetitions between HIP events; GB/s = (bytes read + bytes written) / time.  The
    denominator SURVEY 8(d) / BASELINE.md 3 prescribe next to the datasheet's 8 TB/s."""
    key = (dev.index, gib, reps)
    if key not in _COPY_PEAK:
        n = gib * (1 << 30) // 4
        src = torch.empty((n,), dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        best = 0.0
        # two copies of the same bytes: the runtime's device-to-device copy (a blit kernel) and a vectorised
        # elementwise kernel (read x, write x * 1): the faster of the two is the rate this device streams at
        for op in (lambda: dst.copy_(src), lambda: torch.mul(src, 1.0, out=dst)):
            for _ in range(3):
                op()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(reps):
                op()
            e1.record()
            torch.cuda.synchronize(dev)
            best = max(best, 2.0 * n * 4 * reps / 1e9 / (e0.elapsed_time(e1) / 1e3))
        _COPY_PEAK[key] = best
        del src, dst
        torch.cuda.empty_cache()
    return _COPY_PEAK[key]


def roofline_obj(workload, nbytes, dev_ms, kname, dev=None):
    ach = nbytes / 1e9 / (dev_ms / 1e3)
    traffic, source = None, None
    try:  # PMC bytes per launch from the last committed rocprofv3 counter passes (NOT measured in this run)
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            traffic = json.load(f).get(workload, {}).get("traffic_bytes")
            source = ("profiles/hbm_traffic.json (committed rocprofv3 --pmc passes, not live; the timed loop re-reads one "
                      "67 MB map, which stays in the 256 MB Infinity Cache: the counted reads beyond the L2 are fabric / "
                      "MALL reads, an upper bound of the HBM reads)")
    except OSError:
        pass
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
           "traffic": traffic, "traffic_source": source, "kernel": kname, "kernel_ms": dev_ms,
           "algorithmic_bytes": nbytes}
    if dev is not None:
        try:
            out["peak_measured"] = measured_copy_peak(dev)
            out["peak_measured_how"] = ("device-to-device copy of 1 GiB, 20 repetitions, (read + write) bytes / HIP-event time, this run; "
                                        "the faster of the runtime's copy and an elementwise x * 1 kernel")
            out["frac_of_measured"] = ach / out["peak_measured"]
        except RuntimeError as e:       # (out of memory on a crowded device: the datasheet fraction stands alone)
            out["peak_measured"] = None
            out["peak_measured_how"] = "failed: %s" % str(e)[:120]
    return out


def cold_forward_ms(dev, R, steps=60, warmup=6, n_maps=6):
    """The forward of the roofline leg with the map EVICTED between launches: `n_maps` distinct 67 MB maps (402 MB, more
    than the 256 MB Infinity Cache) visited round-robin, so every launch reads its map from HBM; same RoIs, same output
    buffer.  -> (ms per launch, what was done)"""
    from jdet_amd import _lib as L
    lib = L.lib()
    d = make_inputs("roi_align_rotated", R, 1000, dev)
    rois = d["rois"]
    maps = [d["feat"]] + [torch.randn((1, 256, 256, 256), device=dev).contiguous(memory_format=torch.channels_last)
                          for _ in range(n_maps - 1)]
    out = torch.empty((R, 256, 7, 7), device=dev, memory_format=torch.channels_last)
    wsb = lib.jdet_roi_align_forward_cl_workspace(R, 7, 7)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    ptrs = [m.data_ptr() for m in maps]
    rp, op, wp = rois.data_ptr(), out.data_ptr(), ws.data_ptr()
    st = L.stream_ptr(rois)
    i = [0]

    def step():
        L.check(lib.jdet_roi_align_forward_cl(0, ptrs[i[0] % n_maps], 1, 256, 256, 256, rp, R, 7, 7, 0.25, 2, 1, op, wp,
                                              wsb, st), "fwd_cl")
        i[0] += 1
    _, ms = timed(step, steps, warmup, None, dev)
    del maps
    return ms, ("%d distinct 67 MB maps round-robin (%.0f MB > the 256 MB Infinity Cache): every launch reads its map from "
                "HBM; %d launches between HIP events" % (n_maps, n_maps * 67.1, steps))


def live_traffic(timeout_s=240):
    """HBM-side bytes per launch of the roofline kernel from two rocprofv3 --pmc passes of this file's roi_align_rotated
    workload, run NOW in subprocesses (FETCH_SIZE and WRITE_SIZE need separate passes; FETCH_SIZE counts 128-byte requests
    at 64 bytes on gfx950: doubled, MI355X_MICROARCH.md).  -> (bytes | None, how)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="jdet_pmc_")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            outd = os.path.join(tmp, c)
            cmd = [exe, "--pmc", c, "-f", "csv", "-d", outd, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", "roi_align_rotated", "--no-cpu-baseline", "--no-live-traffic", "--steps", "5", "--warmup", "2"]
            env = dict(os.environ, TMPDIR=tmp)
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            got = []
            for f in glob.glob(os.path.join(outd, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "roi_align_fwd_merged_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, "rocprofv3 --pmc %s gave no rows (exit %d)" % (c, r.returncode)
            vals[c] = sum(got) / len(got)
        traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        return traffic, ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, run by this bench in "
                         "subprocesses), FETCH_SIZE %.0f KB x 2 (gfx950: 128-byte requests counted at 64) + WRITE_SIZE %.0f KB, "
                         "means over the forward kernel's dispatches; the loop re-reads one 67 MB map that stays in the 256 MB "
                         "Infinity Cache: the reads are fabric / MALL reads, an upper bound of the HBM reads"
                         % (vals["FETCH_SIZE"], vals["WRITE_SIZE"]))
    except Exception as e:      # noqa: BLE001 -- reported in the line
        return None, "live counter passes failed: %s: %s" % (type(e).__name__, str(e)[:160])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_extras(a, dev, line, live):
    """the rest of the roofline leg: cold-map forward, live traffic, the planned backward and the forward + backward pair,
    and the reference's true CPU sources (rotated IoU, rotated NMS: kind "reference") beside their device kernels"""
    rf = line["roofline"]
    try:
        ms, how = cold_forward_ms(dev, a.rois)
        rf["kernel_ms_cold"] = ms
        rf["achieved_cold"] = rf["algorithmic_bytes"] / 1e9 / (ms / 1e3)
        rf["frac_cold"] = rf["achieved_cold"] / HBM_PEAK_GBPS
        if rf.get("peak_measured"):
            rf["frac_of_measured_cold"] = rf["achieved_cold"] / rf["peak_measured"]
        rf["cold_how"] = how
    except Exception as e:      # noqa: BLE001
        rf["cold_how"] = "failed: %s: %s" % (type(e).__name__, str(e)[:160])
    if live:
        t, how = live_traffic()
        if t is not None:
            rf["traffic"], rf["traffic_source"] = t, how
        else:
            rf["traffic_source"] = "%s; fallback: %s" % (how, rf.get("traffic_source"))
    also = {}
    for wl, steps in (("roi_align_rotated_bwd", 100), ("roi_align_rotated_bwd_planned", 100), ("roi_align_rotated_pair", 100)):
        try:
            d = make_inputs(wl, a.rois, 1000, dev)
            step, _, _, nbytes, kname, _ = make_step(wl, d)
            _, ms = timed(step, steps, 10, None, dev)
            also[wl] = {"kernel_ms": ms, "algorithmic_bytes": nbytes, "achieved": nbytes / 1e9 / (ms / 1e3),
                        "frac": nbytes / 1e9 / (ms / 1e3) / HBM_PEAK_GBPS, "kernel": kname}
            del step, d
        except Exception as e:      # noqa: BLE001
            also[wl] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    rf["also"] = also
    if not a.no_cpu_baseline:
        ref = {}
        for wl, R, steps in (("box_iou_rotated", a.rois, 50), ("nms_rotated", 2000, 30)):
            try:
                d = make_inputs(wl, R, 1000, dev)
                step, units, unit_name, _, kname, _ = make_step(wl, d)
                t, _ = timed(step, steps, 5, None, dev)
                cb = cpu_baseline(wl, d, R)
                cb["device"] = {"value": units * steps / t, "unit": unit_name + "/s", "kernel": kname}
                ref[wl] = cb
                del step, d
            except Exception as e:      # noqa: BLE001
                ref[wl] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
        line["cpu_baseline_reference"] = ref


def secondary_lines(a, dev):
    """ms/step of the other model configurations of BASELINE.json -- configs[1] RetinaNet-OBB inference, configs[3]
    Oriented R-CNN train step, configs[4] RoI-Transformer R101 train step (4 images) -- 10 timed steps each after the
    headline run, so that the driver's own run witnesses them; an entry that fails reports the error, not the line."""
    out = {}
    for wl, batch, steps, warmup in (("retinanet_infer", 1, 10, 5), ("orcnn_train", 2, 10, 6),
                                     ("roitrans_train", 4, 10, 6)):
        b = argparse.Namespace(**vars(a))
        b.workload, b.batch, b.steps, b.warmup = wl, batch, steps, warmup
        try:
            t0 = time.perf_counter()
            step, keep = make_train(b, 0, dev) if wl in TRAIN_WORKLOADS else make_retinanet_infer(b, 0, dev)
            t, _ = timed(step, steps, warmup, None, dev)
            out[wl] = {"ms_per_step": 1e3 * t / steps, "img_per_s": batch * steps / t, "batch": batch, "steps": steps,
                       "warmup": warmup, "tile": "%dx%d" % (a.size, a.size), "dtype": "f32",
                       "setup_and_run_s": round(time.perf_counter() - t0, 1)}
            del step, keep
        except Exception as e:      # noqa: BLE001 -- reported in the line
            out[wl] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def launch_command(argv, gpus, port=None):
    """`python bench.py --gpus N ...` started by hand (no WORLD_SIZE in the environment): the command that runs the
    same arguments as N ranks of one node, one rank per GPU, rendezvous on 127.0.0.1."""
    import socket
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def selftest_cpu(a, world, rank):
    """launcher / timing plumbing on CPU ranks (gloo): NOT a benchmark; used by tests/test_abi.py to drive
    `bench.py --gpus 2` end to end without a GPU"""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    x = torch.randn(64, 64)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        x = torch.tanh(x @ x.t() / 64)
    t = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
        n = dist.get_world_size()
        dist.destroy_process_group()
    else:
        n = 1
    if rank == 0:
        print(json.dumps({"metric": "selftest steps/s (CPU, launcher plumbing only)", "value": a.steps * n / t,
                          "unit": "step/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": 1e3 * t / a.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "selftest_cpu", "parallelism": "replicas x%d" % n}}))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become N ranks (one per GPU) of one node
        import subprocess
        raise SystemExit(subprocess.call(launch_command(sys.argv[1:], a.gpus)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: reporting the launched world size" % (a.gpus, world),
              file=sys.stderr)
    if a.workload == "selftest_cpu":
        return selftest_cpu(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("JDET_BENCH_FORCE_DIST", "0") == "1"   # exercise the RCCL / DDP path on 1 GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    if a.workload in TRAIN_WORKLOADS or a.workload == "retinanet_infer":
        if a.workload in TRAIN_WORKLOADS:
            step, runner = make_train(a, rank, dev)
            per_step = a.batch
            title = "img/s %s train step, %dx%d synthetic tiles" % (TRAIN_WORKLOADS[a.workload][1], a.size, a.size)
        else:
            step, runner = make_retinanet_infer(a, rank, dev)
            per_step = 1
            title = "img/s RetinaNet-OBB R50-FPN inference (incl. rotated NMS), %dx%d synthetic tile" % (a.size, a.size)
        t, dev_ms = timed(step, a.steps, a.warmup, dist, dev)
        if rank == 0:
            line = {
                "metric": title,
                "value": per_step * world * a.steps / t, "unit": "img/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": 1e3 * t / a.steps, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": {"none": "f32", "bf16": "bf16", "fp16": "f16"}[a.amp],
                "data": "synthetic",
                "config": ({"workload": a.workload,
                            "model": "%s (%s)" % TRAIN_WORKLOADS[a.workload][1:],
                            "global_batch": a.batch * world, "tile": "%dx%d" % (a.size, a.size), "gts_per_image": 64,
                            "optimizer": "SGD mom 0.9 wd 1e-4 clip 35 + StepLR warm-up (the config's lr)",
                            "parallelism": "dp%d (DDP, RCCL all-reduce)" % world} if a.workload in TRAIN_WORKLOADS else
                           {"workload": "retinanet_infer",
                            "model": "RetinaNet-OBB R50-FPN (configs/rotated_retinanet/rotated_retinanet_obb_r50_fpn_1x_dota.py)",
                            "global_batch": world, "tile": "%dx%d" % (a.size, a.size),
                            "post": "top-2000/level -> decode -> multiclass rotated NMS (thr 0.1) -> polys",
                            "parallelism": "replicas x%d (no collective)" % world}),
            }
            # roofline leg: the path's HBM-bound hand-written kernel at the north-star point, measured live
            d = make_inputs("roi_align_rotated", a.rois, 1000, dev)
            rstep, _, _, nbytes, kname, _ = make_step("roi_align_rotated", d)
            _, rms = timed(rstep, 200, 20, None, dev)
            line["roofline"] = roofline_obj("roi_align_rotated", nbytes, rms, kname, dev)
            if world == 1 and not a.no_cpu_baseline:
                cb = cpu_baseline("roi_align_rotated", d, a.rois)
                cb["sample"] = "rotated RoIAlign forward leg (the roofline kernel), not the whole train step: " + cb["sample"]
                line["cpu_baseline"] = cb
            if world == 1:
                del rstep
                roofline_extras(a, dev, line, live=not a.no_live_traffic)
                rstep = None
            if world == 1 and a.workload == "s2anet_train" and a.amp == "none" and not a.no_secondary:
                del step, runner, rstep, d
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                line["secondary"] = secondary_lines(a, dev)
            print(json.dumps(line))
    else:
        d = make_inputs(a.workload, a.rois, 1000 + rank, dev)
        step, units, unit_name, nbytes, kname, dtype = make_step(a.workload, d)
        t, dev_ms = timed(step, a.steps, a.warmup, dist, dev)
        if os.environ.get("JDET_BENCH_CHECKSUM", "0") == "1" and torch.is_tensor(d.get("out")):
            o = d["out"].double()     # A/B runs of one workload must agree: printed to stderr, not part of the line
            print("checksum %s sum %.9e abs %.9e" % (a.workload, float(o.sum()), float(o.abs().sum())), file=sys.stderr)
        if rank == 0:
            line = {
                "metric": ("rotated RoIAlign forward algorithmic GB/s (1024x1024 tile, %d RoIs)" % a.rois
                           if a.workload == "roi_align_rotated" else a.workload + " " + unit_name + "/s"),
                "value": units * a.steps * world / t, "unit": unit_name + "/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": 1e3 * t / a.steps, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                "config": {"workload": a.workload, "fmap": "1x256x256x256 fp32 NHWC", "rois": a.rois,
                           "pooled": "7x7", "sampling_ratio": 2, "spatial_scale": 0.25,
                           "parallelism": "image-parallel x%d (no collective)" % world},
            }
            line["roofline"] = roofline_obj(a.workload, nbytes, dev_ms, kname, dev)
            if world == 1 and not a.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(a.workload, d, a.rois)
            if world == 1 and a.workload == "roi_align_rotated" and not a.no_live_traffic and \
                    os.environ.get("JDET_ROI_FWD_PATH", "roi_cl") == "roi_cl":
                roofline_extras(a, dev, line, live=True)
            print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
