"""GPU: the BASELINE.json configurations at their real workload size (1024 x 1024 tiles), each built through the
registry from the reference's config shape: configs[1] RetinaNet-OBB R50-FPN inference incl. rotated NMS,
configs[2] S2ANet train step, configs[3] Oriented R-CNN train step (2 tiles per GPU), configs[4] RoI-Transformer
with the Resnet101 backbone (2 tiles).  Checks are size-independent properties: finite losses and gradients on
every trainable parameter, the loss keys of the reference heads, the output contract of inference, and a second
step that lowers the loss on the same batch.  (Numerical parity of every operator on the path is what the operator
tests establish; Jittor itself cannot run here to give end-to-end reference numbers)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZE = 1024


def _runner(cfg, dev):
    from jdet_amd.runner import Runner
    torch.manual_seed(0)
    return Runner(cfg, device=dev, conv_autotune=False)   # no solver search in tests


def _train_two_steps(cfg, dev, batch, keys):
    from jdet_amd.runner import synthetic_batch
    r = _runner(cfg, dev)
    images, targets = synthetic_batch(batch, SIZE, dev, seed=11, num_gts=64)
    images = images.contiguous(memory_format=torch.channels_last)
    l0, parts = r.train_step(images, targets)
    assert set(parts) >= set(keys), (sorted(parts), keys)
    assert torch.isfinite(l0) and float(l0) > 0
    for n, p in r.model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, n
            assert torch.isfinite(p.grad).all(), n
    last = l0
    for _ in range(3):
        last, _ = r.train_step(images, targets)
    assert torch.isfinite(last) and float(last) < float(l0)
    return r


def test_cfg1_retinanet_obb_inference_1024(dev):
    """configs[1]: 1 x 3 x 1024 x 1024 tile -> top-2000 / level -> decode -> multiclass rotated NMS -> polys"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.config.named import RETINANET_CFG
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    torch.manual_seed(1)
    m = build_from_cfg(RETINANET_CFG["model"], MODELS).to(dev).eval()
    images, targets = synthetic_batch(1, SIZE, dev, seed=1, num_gts=64)
    with torch.no_grad():
        m.bbox_head.retina_cls.bias.fill_(-2.5)   # random weights: lift the prior so that rotated NMS sees candidates
        res = m(images.contiguous(memory_format=torch.channels_last), targets)
    assert len(res) == 1
    polys, scores, labels = res[0]
    assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]
    assert 0 < polys.shape[0] <= 2000
    assert float(scores.min()) > 0.05 and torch.all(scores[1:] <= scores[:-1])
    assert int(labels.min()) >= 0 and int(labels.max()) <= 14
    # survivors of one class do not violate the NMS rule among themselves
    from jdet_amd.data.np_boxes import poly_to_rotated_box_np
    from jdet_amd.ops.box_iou_rotated import box_iou_rotated
    c = int(labels[0])
    sel = (labels == c).nonzero()[:, 0][:300]
    rb = torch.from_numpy(poly_to_rotated_box_np(polys[sel].cpu().numpy())).to(dev)
    iou = box_iou_rotated(rb, rb)
    assert float(torch.triu(iou, diagonal=1).max()) <= 0.1 + 1e-4


def test_cfg2_s2anet_train_1024(dev):
    from jdet_amd.config.named import S2ANET_CFG
    _train_two_steps(S2ANET_CFG, dev, 2, {"loss_fam_cls", "loss_fam_bbox", "loss_odm_cls", "loss_odm_bbox"})


def test_cfg3_oriented_rcnn_train_1024(dev):
    """configs[3]: batch 16 over 8 GPUs = 2 tiles of 1024 x 1024 per GPU"""
    from jdet_amd.config.named import ORCNN_CFG
    _train_two_steps(ORCNN_CFG, dev, 2, {"loss_cls", "orcnn_bbox_loss", "loss_rpn_cls", "loss_rpn_bbox"})


def test_cfg4_roi_transformer_r101_train_1024(dev):
    """configs[4]: RoI-Transformer with Resnet101 (3-4-23-3), 1024 x 1024 tiles, batch 32 over 8 GPUs = 4 tiles per
    GPU -- the per-GPU batch of the config"""
    from jdet_amd.config.named import roitrans_train_cfg
    r = _train_two_steps(roitrans_train_cfg("Resnet101"), dev, 4,
                         {"loss_rpn_cls", "loss_rpn_bbox", "s0.rbbox_loss_cls", "s0.rbbox_loss_bbox",
                          "s1.rbbox_loss_cls", "s1.rbbox_loss_bbox"})
    assert len(r.model.backbone.layer3) == 23
