"""Feature refinement (R3Det).  Mirrors python/jdet/ops/fr.py: `feature_refine` / `FeatureRefineFunction` (L244-262),
`FR` (L263-276), `FeatureRefineModule` (L277-341).  Kernels: csrc/feature_refine.hip (channels-last, one wave per
location forward; sorted-gather backward instead of 1 + 4 * points float atomics per scalar)."""
import torch
from torch import nn

from jdet_amd.models.utils.weight_init import normal_init

from .. import _lib as L

__all__ = ["feature_refine", "FeatureRefineFunction", "FR", "FeatureRefineModule"]


class FeatureRefineFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, best_rbboxes, spatial_scale, points=1):
        assert points in [1, 5]
        L.need_device(features, best_rbboxes)
        N, C, H, W = features.shape
        x = L.f32c(features.permute(0, 2, 3, 1))                  # NHWC memory; free for channels-last tensors
        boxes = L.f32c(best_rbboxes).reshape(N, H, W, 5)
        out = torch.empty_like(x)
        L.check(L.lib().jdet_feature_refine_forward(L.ptr(x), L.ptr(boxes), N, C, H, W, float(spatial_scale),
                                                    int(points), L.ptr(out), L.stream_ptr(x)),
                "jdet_feature_refine_forward")
        ctx.save_for_backward(boxes)
        ctx.cfg = (N, C, H, W, float(spatial_scale), int(points))
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output):
        (boxes,) = ctx.saved_tensors
        N, C, H, W, scale, points = ctx.cfg
        g = L.f32c(grad_output.permute(0, 2, 3, 1))
        gin = torch.empty_like(g)
        wsb = L.lib().jdet_feature_refine_backward_workspace(N, C, H, W, points)
        ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=g.device)
        L.check(L.lib().jdet_feature_refine_backward(L.ptr(g), L.ptr(boxes), N, C, H, W, scale, points, L.ptr(gin),
                                                     L.ptr(ws), wsb, L.stream_ptr(g)), "jdet_feature_refine_backward")
        return gin.permute(0, 3, 1, 2), None, None, None


feature_refine = FeatureRefineFunction.apply


class FR(nn.Module):
    def __init__(self, spatial_scale, points=1):
        super().__init__()
        self.spatial_scale = float(spatial_scale)
        self.points = points

    def forward(self, features, best_rbboxes):
        return feature_refine(features, best_rbboxes, self.spatial_scale, self.points)

    execute = forward

    def __repr__(self):
        return self.__class__.__name__ + "(spatial_scale={}, points={})".format(self.spatial_scale, self.points)


class FeatureRefineModule(nn.Module):
    """x + FR(conv_5_1(conv_1_5(x)) + conv_1_1(x)) per pyramid level (L277-341)"""

    def __init__(self, in_channels, featmap_strides, conv_cfg=None, norm_cfg=None):
        super().__init__()
        self.in_channels = in_channels
        self.featmap_strides = featmap_strides
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.fr = nn.ModuleList([FR(spatial_scale=1 / s) for s in self.featmap_strides])
        self.conv_5_1 = nn.Conv2d(in_channels, in_channels, kernel_size=(5, 1), stride=1, padding=(2, 0))
        self.conv_1_5 = nn.Conv2d(in_channels, in_channels, kernel_size=(1, 5), stride=1, padding=(0, 2))
        self.conv_1_1 = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.init_weights()

    def init_weights(self):
        normal_init(self.conv_5_1, std=0.01)
        normal_init(self.conv_1_5, std=0.01)
        normal_init(self.conv_1_1, std=0.01)

    def forward(self, x, best_rbboxes):
        """x: list of (N,C,H,W) per level; best_rbboxes: per image a list of (H*W, 5) per level"""
        mlvl_rbboxes = [torch.cat(best_rbbox) for best_rbbox in zip(*best_rbboxes)]
        out = []
        for x_scale, boxes_scale, fr_scale in zip(x, mlvl_rbboxes, self.fr):
            feat_scale = self.conv_5_1(self.conv_1_5(x_scale)) + self.conv_1_1(x_scale)
            out.append(x_scale + fr_scale(feat_scale, boxes_scale))
        return out

    execute = forward
