"""CPU: closed-form pins of the numpy box oracle (SURVEY 8c list) and host logic (registry, config)."""
import math
import os

import numpy as np
import pytest

from oracle import box_oracle as B


def test_norm_angle_range_and_negative_mod():
    a = np.asarray([-10, -3.2, -math.pi / 4, -0.1, 0, 2.3, 3 * math.pi / 4 - 1e-4, 3 * math.pi / 4, 9.7], np.float32)
    r = B.norm_angle(a)
    assert np.all(r >= -math.pi / 4 - 1e-6) and np.all(r < 3 * math.pi / 4 + 1e-6)
    assert np.allclose(np.sin(2 * (r - a)), 0, atol=1e-4)  # differs by a multiple of pi
    assert abs(float(B.norm_angle(np.float32(3 * math.pi / 4))) + math.pi / 4) < 1e-5  # wraps to the low end


def test_codec_round_trip():
    rng = np.random.default_rng(0)
    from tests import inputs as I
    p, g = I.random_obbs(rng, 500), I.random_obbs(rng, 500)
    d = B.bbox2delta_rotated(p, g)
    back = B.delta2bbox_rotated(p, d, wh_ratio_clip=1e-6)
    np.testing.assert_allclose(back[:, :4], g[:, :4], rtol=2e-4, atol=2e-2)
    np.testing.assert_allclose(back[:, 4], B.norm_angle(g[:, 4]), atol=1e-4)
    # means / stds are applied as (d - m) / s and inverted on decode
    m, s = (0.1, -0.2, 0.3, 0.0, 0.05), (0.1, 0.2, 0.2, 0.1, 0.5)
    np.testing.assert_allclose(B.delta2bbox_rotated(p, B.bbox2delta_rotated(p, g, m, s), m, s, 1e-6)[:, :4], g[:, :4],
                               rtol=2e-4, atol=2e-2)
    # multi-class layout (N, 5*K): class k decodes independently
    d2 = np.concatenate([d, d * 0], 1)
    out = B.delta2bbox_rotated(p, d2, wh_ratio_clip=1e-6)
    np.testing.assert_allclose(out[:, :5], back, atol=1e-6)
    np.testing.assert_allclose(out[:, 5:9], p[:, :4], rtol=1e-6)
    # wh_ratio_clip clamps dw, dh
    big = np.asarray([[0, 0, 50.0, -50.0, 0]], np.float32)
    o = B.delta2bbox_rotated(p[:1], big)
    assert abs(o[0, 2] / p[0, 2] - 1000 / 16) < 1e-2 and abs(o[0, 3] / p[0, 3] - 16 / 1000) < 1e-6


def test_assigner_branches():
    # K=3 gts x A=6 anchors, hand-built (assigner.py:L186-209)
    ov = np.asarray([[0.9, 0.45, 0.3, 0.0, 0.2, 0.6],
                     [0.1, 0.45, 0.6, 0.0, 0.2, 0.6],
                     [0.0, 0.10, 0.1, 0.0, 0.2, 0.1]], np.float32)
    gl = np.asarray([7, 8, 9], np.int32)
    gi, mo, lab = B.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, True, gl, 0)
    # anchor0 -> gt1 (pos); anchor1: 0.45 in [0.4,0.5) -> -1; anchor2 -> gt2; anchor3: max 0 -> neg;
    # anchor4: 0.2 is gt3's row max (low-quality, all ties) -> gt3; anchor5: tie 0.6/0.6 -> argmax first (gt1),
    # then low-quality: row0 max is 0.9 (anchor0 only), row1 max 0.6 at anchors 2 and 5 -> gt2 overwrites
    assert gi.tolist() == [1, -1, 2, 0, 3, 2]
    assert lab.tolist() == [7, 0, 8, 0, 9, 8]
    np.testing.assert_allclose(mo, [0.9, 0.45, 0.6, 0.0, 0.2, 0.6])
    gi2, _, lab2 = B.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, False, True, gl, -1)
    assert gi2.tolist() == [1, -1, 2, 0, 0, 1] and lab2.tolist() == [7, -1, 8, -1, -1, 7]
    gi3, _, _ = B.assign_wrt_overlaps(ov, 0.5, (0.1, 0.4), 0.0, False, True, None)
    assert gi3.tolist() == [1, -1, 2, -1, 0, 1]           # tuple thr: [lo, hi)
    gi4, _, _ = B.assign_wrt_overlaps(ov, 0.5, 0.4, 0.3, True, True, None)
    assert gi4.tolist() == [1, -1, 2, 0, 0, 2]            # min_pos_iou filters gt3's low-quality match
    gi5, _, _ = B.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, False, None)
    assert gi5.tolist() == [1, -1, 2, 0, 3, 1]            # only the row argmax (first) per gt


def test_grid_anchors_and_alignconv_zero_offset():
    a = B.grid_anchors_s2anet(8, [4], [1.0], (3, 4), 8)
    assert a.shape == (12, 5)
    np.testing.assert_allclose(a[0], [3.5, 3.5, 32, 32, 0])          # base anchor, centre (s-1)/2
    np.testing.assert_allclose(a[1], [11.5, 3.5, 32, 32, 0])         # x fastest
    np.testing.assert_allclose(a[4], [3.5, 11.5, 32, 32, 0])
    # an axis-aligned anchor of size 3*stride centred on the pixel centre gives zero offsets
    fh, fw, s = 4, 5, 8
    yc, xc = np.meshgrid(np.arange(fh), np.arange(fw), indexing="ij")
    anc = np.stack([xc.reshape(-1) * s, yc.reshape(-1) * s, np.full(fh * fw, 3 * s), np.full(fh * fw, 3 * s),
                    np.zeros(fh * fw)], -1).astype(np.float32)
    off = B.align_conv_offsets(anc, (fh, fw), s)
    assert off.shape == (18, fh, fw)
    np.testing.assert_allclose(off, 0, atol=1e-6)
    # rotating by pi/2 maps tap (dx,dy) -> (-dy,dx): offset of tap 0 (dx=-1,dy=-1) becomes (+2 in x, 0 in y)
    anc[:, 4] = math.pi / 2
    off = B.align_conv_offsets(anc, (fh, fw), s)
    np.testing.assert_allclose(off[0], 0, atol=1e-5)   # dy of tap 0
    np.testing.assert_allclose(off[1], 2, atol=1e-5)   # dx of tap 0


def test_registry_and_reference_config_build():
    """the registry names the named configs use resolve, and the reference's S2ANet config builds unchanged"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.utils import registry as R
    for reg, names in ((R.MODELS, ["S2ANet"]), (R.BACKBONES, ["Resnet50", "Resnet101"]), (R.NECKS, ["FPN"]),
                       (R.HEADS, ["S2ANetHead"]), (R.LOSSES, ["FocalLoss", "SmoothL1Loss", "L1Loss", "CrossEntropyLoss",
                                                              "CrossEntropyLossForRcnn"]),
                       (R.BOXES, ["MaxIoUAssigner", "MaxIoUAssignerRbbox", "RandomSampler", "RandomSamplerRotated",
                                  "PseudoSampler", "DeltaXYWHABBoxCoder", "BboxOverlaps2D", "BboxOverlaps2D_v1",
                                  "BboxOverlaps2D_rotated", "BboxOverlaps2D_rotated_v1",
                                  "AnchorGeneratorRotatedS2ANet", "AnchorGeneratorRotatedRetinaNet"])):
        for n in names:
            assert n in reg, n
    with pytest.raises(AssertionError):
        R.BOXES.get("NoSuchThing")
    assert R.build_from_cfg(None, R.BOXES) is None
    cfg_path = "/root/reference/configs/s2anet/s2anet_r50_fpn_1x_dota.py"
    if not os.path.exists(cfg_path):
        pytest.skip("reference configs not present on this box")
    from jdet_amd.config import Config
    c = Config(cfg_path)
    assert c.model.bbox_head.train_cfg.fam_cfg.assigner.type == "MaxIoUAssigner" and c.nothing is None
    m = R.build_from_cfg(c.model, R.MODELS)
    assert type(m).__name__ == "S2ANet"
    sd = m.state_dict()
    for k in ("backbone.layer1.0.conv1.weight", "neck.fpn_convs.2.conv.weight", "bbox_head.or_conv.weight",
              "bbox_head.align_conv.deform_conv.weight", "bbox_head.or_pool.conv.0.weight", "bbox_head.odm_cls.bias"):
        assert k in sd, k
    assert sd["bbox_head.or_conv.weight"].shape == (32, 256, 1, 3, 3)
    assert sd["bbox_head.odm_cls_convs.0.conv.weight"].shape[1] == 32   # orientation-pooled branch
    assert not any(p.requires_grad for n, p in m.named_parameters() if n.startswith("backbone.layer1."))


def test_config_base_and_cover(tmp_path):
    from jdet_amd.config import Config
    (tmp_path / "base.yaml").write_text("a: {x: 1, y: {p: 1, q: 2}}\nb: [1, 2]\nname: base\n")
    (tmp_path / "child.yaml").write_text("_base_: base.yaml\na: {y: {_cover_: true, r: 3}, z: 9}\nb: [3]\n")
    c = Config(str(tmp_path / "child.yaml"))
    assert c.a.x == 1 and dict(c.a.y) == {"r": 3} and c.a.z == 9 and c.b == [3]
    assert c.name == "base" and c.work_dir == "work_dirs/base"
    (tmp_path / "p.py").write_text("_base_ = ['base.yaml']\nmodel = dict(type='X', k=dict(v=1))\n")
    c2 = Config(str(tmp_path / "p.py"))
    assert c2.model.k.v == 1 and c2.a.y.q == 2 and c2.name == "base"


REFERENCE_CONFIGS = [
    ("configs/s2anet/s2anet_r50_fpn_1x_dota.py", "S2ANet", "S2ANetHead"),
    ("configs/rotated_retinanet/rotated_retinanet_obb_r50_fpn_1x_dota.py", "RotatedRetinaNet", "RotatedRetinaHead"),
    ("configs/oriented_rcnn_r50_fpn_1x_dota_with_flip.py", "OrientedRCNN", "OrientedHead"),
    ("configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py", "RoITransformer", "SharedFCBBoxHeadRbbox"),
]


@pytest.mark.parametrize("rel,model_type,head_type", REFERENCE_CONFIGS)
def test_the_four_named_reference_configs_build_unchanged(rel, model_type, head_type):
    """the config files BASELINE.json names load through `Config` as they are and their model / optimizer / scheduler
    sections build through the registries (build container only: the reference tree is not on the GPU box, where
    jdet_amd.config.named carries the same sections as dicts)"""
    path = os.path.join("/root/reference", rel)
    if not os.path.exists(path):
        pytest.skip("reference configs not present on this box")
    import jdet_amd.models  # noqa: F401
    import jdet_amd.optims  # noqa: F401
    from jdet_amd.config import Config
    from jdet_amd.utils import registry as R
    c = Config(path)
    assert c.model.type == model_type
    m = R.build_from_cfg(c.model, R.MODELS)
    assert type(m).__name__ == model_type
    heads = {type(x).__name__ for x in m.modules()}
    assert head_type in heads, sorted(heads)[:20]
    assert sum(p.numel() for p in m.parameters()) > 20e6
    opt = R.build_from_cfg(c.optimizer, R.OPTIMS, params=list(m.parameters()))
    sch = R.build_from_cfg(c.scheduler, R.SCHEDULERS, optimizer=opt)
    assert type(opt).__name__ == "SGD" and type(sch).__name__ == "StepLR"
    # the dict twin used on the GPU box describes the same model
    from jdet_amd.config import named
    twin = {"S2ANet": named.S2ANET_CFG["model"], "RotatedRetinaNet": named.RETINANET_CFG["model"],
            "OrientedRCNN": named.ORCNN_CFG["model"], "RoITransformer": named.roitrans_cfg()}[model_type]
    m2 = R.build_from_cfg(twin, R.MODELS)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in m2.state_dict().items()}
