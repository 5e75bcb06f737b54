"""GPU: the S2ANet detector built from the reference's own config runs a train step and an inference
pass on the HIP path (DeformConv sampling, ARF, rotated IoU, fused assignment / codec, rotated NMS)."""
import os

import numpy as np
import pytest
import torch

from tests import inputs as I

pytestmark = pytest.mark.gpu


def _model(dev):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    cfg = dict(
        type="S2ANet",
        backbone=dict(type="Resnet50", frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                      pretrained=True),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs="on_input", num_outs=5),
        bbox_head=dict(type="S2ANetHead", num_classes=16, in_channels=256, feat_channels=256, stacked_convs=2,
                       with_orconv=True, anchor_ratios=[1.0], anchor_strides=[8, 16, 32, 64, 128], anchor_scales=[4]))
    torch.manual_seed(0)
    return build_from_cfg(cfg, MODELS).to(dev)


def _targets(n, size, dev, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = 12
        out.append(dict(rboxes=torch.from_numpy(I.random_obbs(rng, k, extent=size, wh=(16.0, size / 4))).to(dev),
                        labels=torch.from_numpy(rng.integers(1, 16, k).astype(np.int32)).to(dev),
                        rboxes_ignore=torch.zeros((0, 5), device=dev), img_size=(size, size), scale_factor=1.0,
                        pad_shape=(size, size)))
    return out


def test_train_step_and_inference(dev):
    from jdet_amd.utils.general import parse_losses
    m = _model(dev)
    m.train()
    size = 256
    imgs = torch.randn(2, 3, size, size, device=dev)
    losses = m(imgs, _targets(2, size, dev))
    assert set(losses) == {"loss_fam_cls", "loss_fam_bbox", "loss_odm_cls", "loss_odm_bbox"}
    total, parsed = parse_losses(losses)
    assert torch.isfinite(total) and total.item() > 0
    total.backward()
    g = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    assert all(v is not None and torch.isfinite(v).all() for v in g.values()), [n for n, v in g.items() if v is None]
    assert g["bbox_head.align_conv.deform_conv.weight"].abs().sum() > 0
    assert g["bbox_head.or_conv.weight"].abs().sum() > 0
    assert g["backbone.layer2.0.conv1.weight"].abs().sum() > 0
    assert all(not p.requires_grad for n, p in m.named_parameters() if n.startswith("backbone.layer1"))
    assert all(not p.requires_grad for n, p in m.named_parameters() if n.startswith("bbox_head.or_pool.conv"))
    # focal-loss prior: at init the classification losses are ~ 1.1-1.2 per positive (bias_init_with_prob(0.01))
    assert 0.3 < parsed["loss_odm_cls"].item() < 5.0
    m.eval()
    with torch.no_grad():
        res = m(imgs, _targets(2, size, dev))
    assert len(res) == 2
    for polys, scores, labels in res:
        assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]


def test_runner_train_steps_reduce_loss(dev):
    """Runner (SGD + clip + StepLR warm-up) on a fixed synthetic batch: the loss goes down and stays finite."""
    from jdet_amd.runner import Runner, synthetic_batch
    from jdet_amd.config.named import RETINANET_CFG, S2ANET_CFG  # noqa: F401
    torch.manual_seed(0)
    r = Runner(S2ANET_CFG, device=dev, conv_autotune=False)   # no solver search in tests
    images, targets = synthetic_batch(2, 256, dev, seed=3, num_gts=16)
    first = None
    for i in range(12):
        loss, parts = r.train_step(images, targets)
        assert torch.isfinite(loss)
        first = loss.item() if first is None else first
    assert loss.item() < first
    assert abs(r.optimizer.cur_lr() - 0.0025 * (1 - (1 - 11 / 500) * (1 - 1 / 3))) < 1e-9


def test_graph_step_matches_eager_step(dev):
    """HIP-graph replay of the whole train step (forward, fused targets, losses, backward, clip, SGD with the lr
    as a device scalar) follows the eager trajectory: same losses step by step (fp32 atomics in the conv /
    deform backward make the last digits run-to-run different, hence the tolerance), same lr schedule."""
    from jdet_amd.config.named import S2ANET_CFG
    from jdet_amd.runner import Runner, synthetic_batch
    images, targets = synthetic_batch(2, 256, dev, seed=3, num_gts=16)
    hist = {}
    for mode in (False, True):
        torch.manual_seed(0)
        r = Runner(S2ANET_CFG, device=dev, conv_autotune=False, graph=mode)
        hist[mode] = [float(r.train_step(images, targets)[0]) for _ in range(8)]
        assert abs(r.optimizer.cur_lr() - 0.0025 * (1 - (1 - 7 / 500) * (1 - 1 / 3))) < 1e-9
        if mode:
            assert len(r._graphs) == 1
    e, g = np.array(hist[False]), np.array(hist[True])
    assert np.all(np.isfinite(g)) and g[-1] < g[0]
    np.testing.assert_allclose(g, e, rtol=2e-2)
    # new data of the same shape replays the same graph; a new shape captures another
    images2, targets2 = synthetic_batch(2, 256, dev, seed=9, num_gts=16)
    l2 = float(r.train_step(images2, targets2)[0])
    assert np.isfinite(l2) and len(r._graphs) == 1
    images3, targets3 = synthetic_batch(2, 256, dev, seed=9, num_gts=5)
    assert np.isfinite(float(r.train_step(images3, targets3)[0])) and len(r._graphs) == 2


def test_retinanet_obb_train_and_infer(dev):
    """RetinaNet-OBB (BASELINE configs[1]) built from the reference config shape: train losses finite, 9 anchors
    per location, inference returns (polys, scores, labels)."""
    from jdet_amd.config.named import RETINANET_CFG, S2ANET_CFG  # noqa: F401
    import jdet_amd.models  # noqa: F401
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.general import parse_losses
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    torch.manual_seed(0)
    m = build_from_cfg(RETINANET_CFG["model"], MODELS).to(dev)
    assert m.bbox_head.num_anchors == 9 and m.bbox_head.retina_reg.out_channels == 45
    images, targets = synthetic_batch(2, 256, dev, seed=4, num_gts=8)
    m.train()
    total, parsed = parse_losses(m(images, targets))
    assert set(parsed) == {"loss_cls", "loss_bbox"} and torch.isfinite(total)
    total.backward()
    assert m.bbox_head.retina_cls.weight.grad.abs().sum() > 0
    m.eval()
    with torch.no_grad():
        # lift the classification prior so that some anchors pass score_thr and reach rotated NMS
        m.bbox_head.retina_cls.bias.fill_(-2.0)
        res = m(images, targets)
    assert len(res) == 2
    polys, scores, labels = res[0]
    assert polys.shape[0] > 0 and polys.shape[1] == 8 and scores.min() > 0.05 and labels.max() < 15
    assert torch.all(scores[1:] <= scores[:-1])


def test_packed_small_levels_equal_the_per_level_path(dev):
    """S2ANetHead runs the conv towers of the small pyramid levels on one packed tensor (LevelPack: levels placed
    side by side, zero gaps kept zero); outputs and gradients equal the per-level execution up to the conv library's
    accumulation order"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.models.roi_heads.s2anet_head import S2ANetHead
    from jdet_amd.models.utils.level_pack import LevelPack
    torch.manual_seed(0)
    head = S2ANetHead(num_classes=16, in_channels=32, feat_channels=32, stacked_convs=2, with_orconv=True,
                      anchor_strides=[8, 16, 32, 64, 128]).to(dev)
    head.train()
    sizes = [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)]          # 1600 positions: stays per level; 4 packed
    feats = [torch.randn(2, 32, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for h, w in sizes]
    pack = LevelPack(sizes[1:], dev)
    # placement: largest level top left, the others in a side column; any two levels at least one row / column apart,
    # tighter than stacking along H (41 x 20)
    import itertools
    assert pack.height * pack.width < 41 * 20 and pack.mask.sum().item() == sum(h * w for h, w in sizes[1:])
    for (sa, pa), (sb, pb) in itertools.combinations(list(zip(pack.sizes, pack.places)), 2):
        assert pa[0] + sa[0] < pb[0] or pb[0] + sb[0] < pa[0] or pa[1] + sa[1] < pb[1] or pb[1] + sb[1] < pa[1]
    again = pack.unpack(pack.pack([f.detach() for f in feats[1:]]))
    assert all(torch.equal(a, f.detach()) for a, f in zip(again, feats[1:]))

    def run(limit):
        head.pack_max_positions = limit
        for f in feats:
            f.grad = None
        head.zero_grad()
        outs = head._level_outputs(feats)
        total = sum((o.float() ** 2).sum() * (0.5 + k) for k, group in enumerate(outs) for o in group
                    if o is not None and o.requires_grad)
        total.backward()
        flat = [o.detach().clone() for group in outs for o in group if o is not None]
        grads = [f.grad.clone() for f in feats] + [p.grad.clone() for p in head.parameters() if p.grad is not None]
        return flat, grads

    ref_out, ref_grad = run(0)
    got_out, got_grad = run(1024)
    assert len(ref_out) == len(got_out) and len(ref_grad) == len(got_grad)
    for a, b in zip(got_out + got_grad, ref_out + ref_grad):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4 * max(1.0, float(b.abs().max())))


def test_side_stream_branch_gives_the_single_stream_gradients(dev, monkeypatch):
    """advisor finding (round 5): the packed levels run on a side stream under parameter ALIASES (a private torch helper);
    outputs and every gradient must equal the single-stream execution (JDET_HEAD_STREAMS=0), and a torch without the
    helper takes the single-stream path instead of failing"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.models.roi_heads import s2anet_head as SH
    torch.manual_seed(1)
    head = SH.S2ANetHead(num_classes=16, in_channels=32, feat_channels=32, stacked_convs=2, with_orconv=True,
                         anchor_strides=[8, 16, 32, 64, 128]).to(dev)
    head.train()
    head.pack_max_positions = 1024
    sizes = [(40, 40), (20, 20), (10, 10), (5, 5), (3, 3)]
    feats = [torch.randn(2, 32, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for h, w in sizes]

    def run():
        for f in feats:
            f.grad = None
        head.zero_grad()
        outs = head._level_outputs(feats)
        total = sum((o.float() ** 2).sum() * (0.5 + k) for k, group in enumerate(outs) for o in group
                    if o is not None and o.requires_grad)
        total.backward()
        torch.cuda.synchronize()
        return ([o.detach().clone() for group in outs for o in group if o is not None],
                [f.grad.clone() for f in feats] + [p.grad.clone() for p in head.parameters() if p.grad is not None])

    monkeypatch.setattr(SH, "HEAD_STREAMS", True)
    SH._SIDE.clear()
    a = run()
    assert len(SH._SIDE) == 1, "the packed levels did not take the side stream"
    monkeypatch.setattr(SH, "HEAD_STREAMS", False)
    b = run()
    monkeypatch.setattr(SH, "HEAD_STREAMS", True)
    monkeypatch.setattr(SH, "_reparametrize_module", None)      # a torch without the private helper
    SH._SIDE.clear()
    c = run()
    assert len(SH._SIDE) == 0
    for other in (a, c):
        for u, v in zip(other[0] + other[1], b[0] + b[1]):
            torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-5 * max(1.0, float(v.abs().max())))
