"""GPU: FPN-routed RoIAlign (masked multi-level launch) against a per-level oracle, horizontal NMS, and an
Oriented R-CNN train step + inference built from the reference config shape."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu


@pytest.fixture
def reference_order():
    """bit-exact comparisons run the RoIAlign kernels in the reference's operation order"""
    from jdet_amd.ops import _roi_common as RC
    prev = RC.set_arithmetic("reference")
    yield
    RC.set_arithmetic(prev)


def test_oriented_extractor_vs_per_level_oracle(dev, reference_order):
    from jdet_amd.models.roi_extractors import OrientedSingleRoIExtractor, RboxSingleRoIExtractor, SingleRoIExtractor
    rng = np.random.default_rng(0)
    strides = [4, 8, 16, 32]
    feats_np = [rng.standard_normal((2, 8, 256 // s, 256 // s)).astype(np.float32) for s in strides]
    obbs = I.random_obbs(rng, 60, extent=256.0, wh=(8.0, 400.0))
    rois = I.rois_from_obbs(obbs, rng.integers(0, 2, 60))
    feats = [torch.from_numpy(f).to(dev).requires_grad_(True) for f in feats_np]
    ext = OrientedSingleRoIExtractor(dict(type="ROIAlignRotated_v1", output_size=7, sampling_ratio=2), 8, strides,
                                     extend_factor=(1.4, 1.2)).to(dev)
    out = ext(feats, torch.from_numpy(rois).to(dev))
    # oracle: enlarge (h*1.4, w*1.2), level from the enlarged RoI, pool on that level only
    r2 = rois.copy()
    r2[:, 3] *= 1.2
    r2[:, 4] *= 1.4
    lvl = np.clip(np.floor(np.log2(np.sqrt(r2[:, 3] * r2[:, 4]) / 56 + 1e-6)), 0, 3).astype(int)
    ref = np.zeros((60, 8, 7, 7), np.float32)
    for i, s in enumerate(strides):
        m = lvl == i
        if m.any():
            ref[m] = O.roi_align_forward(O.V_ROT_V1, feats_np[i], r2[m], (7, 7), 1.0 / s, 2)
    assert len(set(lvl.tolist())) >= 3
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    for i, s in enumerate(strides):
        m = lvl == i
        gref = O.roi_align_backward(O.V_ROT_V1, g[m], r2[m], feats_np[i].shape, 1.0 / s, 2) if m.any() else 0 * feats_np[i]
        np.testing.assert_allclose(feats[i].grad.cpu().contiguous().numpy(), gref, atol=3e-5)
    # the other two extractors resolve their layer class on the right ops module and route the same way
    e2 = RboxSingleRoIExtractor(dict(type="ROIAlignRotated", output_size=7, sampling_ratio=2), 8, strides).to(dev)
    o2 = e2([f.detach() for f in feats], torch.from_numpy(rois).to(dev)).cpu().numpy()
    lvl2 = np.clip(np.floor(np.log2(np.sqrt(rois[:, 3] * rois[:, 4]) / 56 + 1e-6)), 0, 3).astype(int)
    for i, s in enumerate(strides):
        m = lvl2 == i
        if m.any():
            np.testing.assert_array_equal(o2[m], O.roi_align_forward(O.V_ROT, feats_np[i], rois[m], (7, 7), 1.0 / s, 2))
    h = I.obb_to_hbb_rois(rois)
    e3 = SingleRoIExtractor(dict(type="ROIAlign", output_size=7, sampling_ratio=2, version=1), 8, strides).to(dev)
    o3 = e3([f.detach() for f in feats], torch.from_numpy(h).to(dev)).cpu().numpy()
    lvl3 = np.clip(np.floor(np.log2(np.sqrt((h[:, 3] - h[:, 1] + 1) * (h[:, 4] - h[:, 2] + 1)) / 56 + 1e-6)), 0, 3).astype(int)
    for i, s in enumerate(strides):
        m = lvl3 == i
        if m.any():
            np.testing.assert_array_equal(o3[m], O.roi_align_forward(O.V_HBB1, feats_np[i], h[m], (7, 7), 1.0 / s, 2))


def test_horizontal_nms(dev):
    from jdet_amd.ops.nms import nms
    rng = np.random.default_rng(1)
    c = rng.uniform(0, 200, (400, 2))
    wh = rng.uniform(10, 60, (400, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = (rng.uniform(0, 1, 400) + np.arange(400) * 1e-7).astype(np.float32)
    keep = nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.5).cpu().numpy()
    # numpy greedy reference: IoU > thr suppresses, result in score order
    order = np.argsort(-scores, kind="stable")
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sup = np.zeros(400, bool)
    ref = []
    for i in order:
        if sup[i]:
            continue
        ref.append(i)
        xx1, yy1 = np.maximum(boxes[i, 0], boxes[:, 0]), np.maximum(boxes[i, 1], boxes[:, 1])
        xx2, yy2 = np.minimum(boxes[i, 2], boxes[:, 2]), np.minimum(boxes[i, 3], boxes[:, 3])
        inter = np.clip(xx2 - xx1, 0, None) * np.clip(yy2 - yy1, 0, None)
        iou = inter / (area[i] + area - inter)
        sup |= iou > 0.5
    assert keep.tolist() == ref
    assert nms(torch.zeros((0, 4), device=dev), torch.zeros((0,), device=dev), 0.5).numel() == 0


def _orcnn(dev):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    cfg = dict(
        type="OrientedRCNN",
        backbone=dict(type="Resnet50", frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                      pretrained=True),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
        rpn=dict(type="OrientedRPNHead", in_channels=256, num_classes=1, nms_pre=2000, nms_post=2000),
        bbox_head=dict(type="OrientedHead", num_classes=15, in_channels=256, fc_out_channels=1024))
    torch.manual_seed(0)
    return build_from_cfg(cfg, MODELS).to(dev)


def test_oriented_rcnn_train_step_and_inference(dev):
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.general import parse_losses
    m = _orcnn(dev)
    m.train()
    images, targets = synthetic_batch(2, 256, dev, seed=5, num_gts=10)
    losses = m(images, targets)
    assert set(losses) == {"loss_cls", "orcnn_bbox_loss", "loss_rpn_cls", "loss_rpn_bbox"}
    total, parsed = parse_losses(losses)
    assert torch.isfinite(total) and total.item() > 0
    total.backward()
    g = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    assert all(v is not None and torch.isfinite(v).all() for v in g.values()), [n for n, v in g.items() if v is None]
    assert g["bbox_head.shared_fcs.0.weight"].abs().sum() > 0 and g["neck.fpn_convs.0.conv.weight"].abs().sum() > 0
    # log(16) ~ 2.77 at init for a 16-way softmax
    assert 1.5 < parsed["loss_cls"].item() < 4.0
    m.eval()
    with torch.no_grad():
        res = m(images, targets)
    assert len(res) == 2
    for polys, scores, labels in res:
        assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]


def test_level_labels_equal_the_coordinate_offset_trick(dev):
    """nms(labels=level ids) keeps exactly what the reference's per-level coordinate offsets keep
    (oriented_rpn_head.py:L214-219)"""
    from jdet_amd.ops.nms import nms
    rng = np.random.default_rng(8)
    n = 3000
    c = rng.uniform(0, 300, (n, 2))
    wh = rng.uniform(10, 80, (n, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)).to(dev)
    scores = torch.from_numpy((rng.uniform(0, 1, n) + np.arange(n) * 1e-7).astype(np.float32)).to(dev)
    ids = torch.from_numpy(rng.integers(0, 5, n)).to(dev)
    off = ids.to(boxes.dtype) * (boxes.max() - boxes.min() + 1)
    a = nms(boxes + off[:, None], scores, 0.7)
    b = nms(boxes, scores, 0.7, labels=ids)
    assert torch.equal(a, b) and 0 < a.numel() < n


def test_train_step_runs_without_host_synchronisation(dev):
    """the Oriented R-CNN train step (RPN targets, proposals, RCNN sampling / targets, RoIAlign, losses,
    backward) has fixed shapes and no device -> host round trip: with PyTorch's sync debug mode set to "error" any
    nonzero / boolean-mask indexing / .item() / bool(tensor) in the step raises."""
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.general import parse_losses
    m = _orcnn(dev)
    m.train()
    images, targets = synthetic_batch(2, 256, dev, seed=5, num_gts=10)
    total, _ = parse_losses(m(images, targets))      # warm-up: anchor caches, MIOpen workspaces
    total.backward()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        total, parsed = parse_losses(m(images, targets))
        total.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.isfinite(total)


def test_proposal_table_contract(dev):
    """OrientedRPNHead hands over `nms_post` rows per image, sorted by score, padding rows with score -1; the kept rows
    are what the reference's dynamic-shape pipeline keeps (same NMS on the same boxes)"""
    from jdet_amd.models.roi_heads import OrientedRPNHead
    from jdet_amd.ops.bbox_transforms import obb2hbb
    from jdet_amd.ops.nms import nms
    torch.manual_seed(3)
    rpn = OrientedRPNHead(in_channels=16, feat_channels=16, nms_pre=300, nms_post=200).to(dev).eval()
    feats = [torch.randn(2, 16, 128 // s, 128 // s, device=dev) for s in (4, 8, 16, 32, 64)]
    targets = [dict(img_size=(128, 128), pad_shape=(128, 128))] * 2
    with torch.no_grad():
        tables, losses = rpn(feats, targets)
    assert losses == {} and len(tables) == 2
    for tab in tables:
        assert tab.shape == (200, 6)
        s = tab[:, 5]
        alive = s >= 0
        n = int(alive.sum())
        assert 0 < n <= 200 and bool(alive[:n].all()) and not bool(alive[n:].any())
        assert bool((s[:n - 1] >= s[1:n]).all())
        # idempotence: the survivors do not suppress each other under the same rule (per level unknown here, so
        # check with the plain rule on all: a subset of what per-level NMS allows may remain -> only sanity bounds)
        keep = nms(obb2hbb(tab[:n, :5]), s[:n], 0.8)
        assert keep.numel() <= n


def test_graph_replays_carry_no_garbage_gradient(dev):
    """Oriented R-CNN, forward + backward captured as a rank of a two-rank job captures it, replayed from identical
    state: every parameter's gradient segment stays within the normal spread of the replays (proposal NMS / sampling:
    ~0.1 of the gradient norm).  Round 5 found the two FC bias gradients of the R-CNN head at 5e4 against 0.2 in EVERY
    replay -- the framework's multi-workgroup reduce clears its semaphores with a memset node that does not survive
    replay on this stack; the heads' FC layers now form that gradient with plain kernels (ops/linear.py)."""
    from jdet_amd.config.named import ORCNN_CFG
    from jdet_amd.runner import Runner, synthetic_batch
    torch.manual_seed(1234)
    r = Runner(ORCNN_CFG, device=dev, conv_autotune=False, graph=True, ddp=False)
    images, targets = synthetic_batch(1, 256, dev, seed=500, num_gts=12)
    images = images.contiguous(memory_format=torch.channels_last)
    for step in range(2):
        torch.manual_seed(9000 + step)
        r.train_step(images, targets)
    r.world_size = 2                       # g1 = forward + backward only (the update is a second graph)
    r.model.train()
    st = r._capture(images, targets)
    r.world_size = 1
    torch.cuda.synchronize()
    segs, off = [], 0
    for n, p in r.model.named_parameters():
        if p.requires_grad:
            segs.append((n, off, off + p.numel()))
            off += p.numel()
    flats = []
    for k in range(6):
        torch.manual_seed(777)
        st["g1"].replay()
        torch.cuda.synchronize()
        flats.append(st["flat"].clone())
    ref = flats[0]
    assert torch.isfinite(ref).all()
    for f in flats[1:]:
        assert float((f - ref).norm()) <= 0.5 * float(ref.norm())
        for n, a, b in segs:
            assert float(f[a:b].norm()) <= 20.0 * float(ref[a:b].norm()) + 1e-3, n
