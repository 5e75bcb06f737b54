"""CPU: the oracle (oracle/jdet_oracle.cpp) against the golden vectors (tests/golden/gen_golden.py) and the reference's
known-answer literals.  IoU / NMS / ARF vectors are outputs of the reference's own CPU sources compiled in the build
container: these tests are what pins those parts of the oracle.  RoIAlign / DeformConv vectors are restatement output
(CUDA-only reference sources; the pins of those parts are the closed forms of tests/test_closed_form_cpu.py): here
they are regression checks."""
import numpy as np
import pytest

from oracle import oracle as O

ROI_CASES = [((7, 7), 2), ((3, 5), 0), ((2, 2), 3)]


@pytest.mark.parametrize("variant,nm", [(O.V_ROT, "rot"), (O.V_ROT_V1, "rot_v1"), (O.V_HBB0, "hbb0"), (O.V_HBB1, "hbb1")])
@pytest.mark.parametrize("hw,s", ROI_CASES)
def test_roi_align_oracle_vs_golden(golden, variant, nm, hw, s):
    g = golden("roi_align")
    rois = g["hrois"] if variant in (O.V_HBB0, O.V_HBB1) else g["rois"]
    key = "%s_%dx%d_s%d" % (nm, hw[0], hw[1], s)
    y = O.roi_align_forward(variant, g["feat"], rois, hw, float(g["scale"]), s)
    np.testing.assert_array_equal(y, g["y_" + key])  # bit-exact
    gi = O.roi_align_backward(variant, g["g_" + key], rois, g["feat"].shape, float(g["scale"]), s)
    np.testing.assert_allclose(gi, g["gi_" + key], rtol=0, atol=1e-5)


@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 0)])
def test_riroi_align_oracle_vs_golden(golden, hw, s):
    g = golden("riroi_align")
    key = "ri_%dx%d_s%d" % (hw[0], hw[1], s)
    nO = int(g["nO"])
    y = O.roi_align_forward(O.V_RI, g["feat"], g["rois"], hw, float(g["scale"]), s, nO)
    np.testing.assert_array_equal(y, g["y_" + key])
    gi = O.roi_align_backward(O.V_RI, g["g_" + key], g["rois"], g["feat"].shape, float(g["scale"]), s, nO)
    np.testing.assert_allclose(gi, g["gi_" + key], rtol=0, atol=1e-5)


def test_roi_align_edge_semantics(golden):
    """first edge RoI is completely outside -> exact zeros; constant map -> exact 1 inside."""
    g = golden("roi_align")
    y = g["y_rot_7x7_s2"]
    n_rand = g["rois"].shape[0] - 10
    assert np.all(y[n_rand] == 0)
    ones = np.ones((1, 2, 16, 16), np.float32)
    roi = np.asarray([[0, 32, 32, 20, 12, 0.4]], np.float32)
    out = O.roi_align_forward(O.V_ROT, ones, roi, (7, 7), 0.25, 2)
    np.testing.assert_allclose(out, 1.0, atol=1e-6)


def test_iou_oracle_vs_golden(golden):
    g = golden("box_iou_rotated")
    np.testing.assert_array_equal(O.box_iou_rotated(g["b1"], g["b2"], 0, 0), g["iou"])
    np.testing.assert_array_equal(O.box_iou_rotated(g["b1"], g["b2"], 1, 0), g["iou_v1"])
    np.testing.assert_array_equal(O.box_iou_rotated(g["b1"], g["b2"], 0, 1), g["iou_cudasort"])
    big = O.box_iou_rotated(g["big1"], g["big2"], 0, 0)
    assert float(big.astype(np.float64).sum()) == float(g["iou_big_sum"])
    assert int((big > 0).sum()) == int(g["iou_big_nnz"])


def test_iou_known_answers(golden):
    g = golden("box_iou_rotated")
    # reference literal box_iou_rotated.py:L513-516; analytic answer [[1,.2],[.2,1]]
    np.testing.assert_allclose(g["iou_lit"], [[1, 0.2], [0.2, 1]], atol=1e-6)
    np.testing.assert_array_equal(O.box_iou_rotated(g["lit"], g["lit"]), g["iou_lit"])
    # analytic: two unit squares offset by half a side, one rotated by 90 deg (same footprint)
    a = np.asarray([[0, 0, 2, 2, 0]], np.float32)
    b = np.asarray([[1, 0, 2, 2, np.pi / 2]], np.float32)
    np.testing.assert_allclose(O.box_iou_rotated(a, b), [[1 / 3]], atol=1e-6)
    # empty inputs
    assert O.box_iou_rotated(np.zeros((0, 5), np.float32), a).shape == (0, 1)


def test_nms_oracle_vs_golden(golden):
    g = golden("nms_rotated")
    order = np.argsort(-g["lit_scores"], kind="stable").astype(np.int32)
    k = O.nms_rotated_keep(g["lit_dets"], order, 0.3)
    assert list(np.nonzero(k)[0]) == [2]  # nms_rotated.py:L599-603 literal
    for nm in "abc":
        dets, scores, labels = g["dets_" + nm], g["scores_" + nm], g["labels_" + nm]
        order = np.argsort(-scores, kind="stable").astype(np.int32)
        d6 = np.concatenate([dets, labels[:, None]], 1)
        for thr in (0.1, 0.5):
            np.testing.assert_array_equal(O.nms_rotated_keep(dets, order, thr), g["keep5_%s_%g" % (nm, thr)])
            np.testing.assert_array_equal(O.nms_rotated_keep(d6, order, thr), g["keep6_%s_%g" % (nm, thr)])


def test_nms_rule_ge_vs_gt():
    """CPU rule `>=` vs CUDA rule `>` differ exactly at iou == thr (nms_rotated.py:L444 vs L403)."""
    dets = np.asarray([[0, 0, 2, 2, 0], [1, 0, 2, 2, 0]], np.float32)  # iou = 1/3
    iou = float(O.box_iou_rotated(dets[:1], dets[1:])[0, 0])
    order = np.asarray([0, 1], np.int32)
    assert list(O.nms_rotated_keep(dets, order, iou, cmp_ge=1)) == [True, False]
    assert list(O.nms_rotated_keep(dets, order, iou, cmp_ge=0)) == [True, True]


def test_dcn_oracle_vs_golden(golden):
    g = golden("deform_conv")
    for nm in "abc":
        k, pad, stride, dil, dg = [int(v) for v in g["cfg_" + nm]]
        a = (k, k, (pad, pad), (stride, stride), (dil, dil), dg)
        im, off = g["im_" + nm], g["off_" + nm]
        np.testing.assert_array_equal(O.deform_im2col(im, off, *a), g["col_" + nm])
        np.testing.assert_allclose(O.deform_col2im(g["gcol_" + nm], off, im.shape, *a), g["gim_" + nm], atol=1e-5)
        np.testing.assert_array_equal(O.deform_col2im_coord(g["gcol_" + nm], im, off, *a), g["goff_" + nm])


def test_dcn_zero_offset_is_plain_im2col():
    rng = np.random.default_rng(0)
    im = rng.standard_normal((1, 2, 5, 6)).astype(np.float32)
    off = np.zeros((1, 18, 5, 6), np.float32)
    col = O.deform_im2col(im, off, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    pad = np.pad(im, ((0, 0), (0, 0), (1, 1), (1, 1)))
    for c in range(2):
        for i in range(3):
            for j in range(3):
                np.testing.assert_array_equal(col[c * 9 + i * 3 + j, 0], pad[0, c, i:i + 5, j:j + 6])


def test_arf_oracle_vs_golden(golden):
    g = golden("arf")
    np.testing.assert_array_equal(O.arf_forward(g["w"], g["idx"]), g["y"])
    np.testing.assert_array_equal(O.arf_backward(g["idx"], g["g"]), g["gw"])
    np.testing.assert_array_equal(O.arf_forward(g["w1"], g["idx1"]), g["y1"])
    # rotation 0 is the identity permutation
    y = g["y"].reshape(4, 8, 3, 8, 3, 3)
    np.testing.assert_array_equal(y[:, 0], g["w"])


def test_dcn_v2_restatement_vs_golden(golden):
    """regression vectors of the round-3 operators (written by gen_golden.py behind the closed-form pins)"""
    g = golden("dcn_v2")
    for nm in "abc":
        k, ph, pw, stride, dil, dg = [int(v) for v in g["cfg_" + nm]]
        a = ((ph, pw), (stride, stride), (dil, dil), dg)
        x, off, mask, w = g["x_" + nm], g["off_" + nm], g["mask_" + nm], g["w_" + nm]
        np.testing.assert_array_equal(O.dcn_v2_forward(x, off, mask, w, g["bias_" + nm], *a), g["y_" + nm])
        for got, key in zip(O.dcn_v2_backward(x, off, mask, w, g["g_" + nm], *a), ("gi", "go", "gm", "gw", "gb")):
            np.testing.assert_array_equal(got, g[key + "_" + nm])
    od, G, P, part, spp = [int(v) for v in g["ps_cfg"]]
    scale, tstd = [float(v) for v in g["ps_f"]]
    for nm, no_trans in (("plain", True), ("deform", False)):
        y, cnt = O.deform_psroi_forward(g["ps_x"], g["ps_rois"], g["ps_trans"], no_trans, scale, od, G, P, part, spp, tstd)
        np.testing.assert_array_equal(y, g["ps_y_" + nm])
        np.testing.assert_array_equal(cnt, g["ps_cnt_" + nm])
        gi, gt = O.deform_psroi_backward(g["ps_g"], cnt, g["ps_x"], g["ps_rois"], g["ps_trans"], no_trans, scale, od,
                                         G, P, part, spp, tstd)
        np.testing.assert_array_equal(gi, g["ps_gi_" + nm])
        if not no_trans:
            np.testing.assert_array_equal(gt, g["ps_gt_" + nm])


def test_convex_restatement_vs_golden(golden):
    g = golden("convex_ops")
    np.testing.assert_array_equal(O.convex_iou(g["pointsets"], g["quads"]), g["ious"])
    np.testing.assert_array_equal(O.min_area_bbox(g["pointsets"]), g["boxes"])
    np.testing.assert_array_equal(O.convex_sort(g["pts"], g["masks"], True), g["sort_circular"])
    np.testing.assert_array_equal(O.convex_sort(g["pts"], g["masks"], False), g["sort_open"])
