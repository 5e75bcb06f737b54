"""GPU: RoI-Transformer (configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py model section, built through the
registry) -- one train step with finite losses / gradients everywhere, inference output contract, and the
stage-1 -> stage-2 hand-off invariants."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


from jdet_amd.config.named import roitrans_cfg  # noqa: E402




def test_roi_transformer_train_step_and_inference(dev):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.general import parse_losses
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    torch.manual_seed(0)
    m = build_from_cfg(roitrans_cfg(), MODELS).to(dev)
    m.train()
    images, targets = synthetic_batch(2, 256, dev, seed=7, num_gts=10)
    losses = m(images, targets)
    assert set(losses) == {"loss_rpn_cls", "loss_rpn_bbox", "s0.rbbox_loss_cls", "s0.rbbox_acc", "s0.rbbox_loss_bbox",
                           "s1.rbbox_loss_cls", "s1.rbbox_acc", "s1.rbbox_loss_bbox"}
    total, parsed = parse_losses(losses)
    assert torch.isfinite(total) and total.item() > 0
    # 16-way softmax at init: ~ log 16 up to the scale of the random backbone's features
    assert 1.0 < parsed["s0.rbbox_loss_cls"].item() < 12.0 and 1.0 < parsed["s1.rbbox_loss_cls"].item() < 12.0
    total.backward()
    g = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    missing = [n for n, v in g.items() if v is None]
    assert not missing, missing
    assert all(torch.isfinite(v).all() for v in g.values())
    for n in ("bbox_head.shared_fcs.0.weight", "rbbox_head.shared_fcs.0.weight", "rpn_head.rpn_conv.weight",
              "neck.fpn_convs.0.conv.weight"):
        assert g[n].abs().sum() > 0, n
    m.eval()
    with torch.no_grad():
        res = m(images[:1], targets[:1])
    assert len(res) == 1
    polys, scores, labels = res[0]
    assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]
    assert polys.shape[0] <= 2000
    if scores.numel():
        assert float(scores.min()) > 0.05 and int(labels.min()) >= 0 and int(labels.max()) <= 14


def test_rpn_proposals_contract(dev):
    """FasterrcnnHead.get_bboxes: (<= max_num, 5) proposals inside the image, scores descending"""
    from jdet_amd.models.roi_heads import FasterrcnnHead
    torch.manual_seed(1)
    rpn = FasterrcnnHead(in_channels=16, feat_channels=16, anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                         anchor_strides=[4, 8, 16, 32, 64],
                         loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=True)).to(dev)
    feats = [torch.randn(2, 16, 256 // s, 256 // s, device=dev) for s in (4, 8, 16, 32, 64)]
    metas = [dict(img_shape=(256, 256), pad_shape=(256, 256), scale_factor=1.0)] * 2
    cfg = dict(nms_across_levels=False, nms_pre=500, nms_post=300, max_num=400, nms_thr=0.7, min_bbox_size=0)
    with torch.no_grad():
        props = rpn.get_bboxes(*rpn(feats), metas, cfg)
    assert len(props) == 2
    for p in props:
        assert p.shape[1] == 5 and 0 < p.shape[0] <= 400
        assert float(p[:, :4].min()) >= 0 and float(p[:, :4].max()) <= 255
        s = p[:, 4].cpu().numpy()
        assert np.all(np.diff(s) <= 1e-7)
