"""Rotated anchor generators of S2ANet / RetinaNet-OBB.  Mirror
python/jdet/models/boxes/anchor_generator.py: `AnchorGeneratorRotatedRetinaNet` L7-110,
`AnchorGeneratorRotatedS2ANet` L112-196.  Base anchor centre (base-1)/2; grid is row-major over
locations with the A base anchors fastest."""
import numpy as np
import torch

from jdet_amd.utils.registry import BOXES


class _RotatedAnchorBase:
    def _finish(self, base_size, scales, ratios, angles, scale_major, ctr):
        self.base_size = base_size
        self.scales = torch.as_tensor(scales, dtype=torch.float32)
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32)
        self.angles = torch.as_tensor(angles, dtype=torch.float32)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()
        self._dev_cache = {}

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        w = h = self.base_size
        if self.ctr is None:
            x_ctr, y_ctr = 0.5 * (w - 1), 0.5 * (h - 1)
        else:
            x_ctr, y_ctr = self.ctr
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        assert self.scale_major, "AnchorGeneratorRotated only support scale-major anchors!"
        ones = torch.ones_like(self.angles)
        ws = (w * w_ratios[:, None, None] * self.scales[None, :, None] * ones[None, None, :]).view(-1)
        hs = (h * h_ratios[:, None, None] * self.scales[None, :, None] * ones[None, None, :]).view(-1)
        angles = self.angles.repeat(len(self.scales) * len(self.ratios))
        xc = x_ctr + torch.zeros_like(ws)
        yc = y_ctr + torch.zeros_like(ws)
        return torch.stack([xc, yc, ws, hs, angles], dim=-1)

    @staticmethod
    def _meshgrid(x, y, row_major=True):
        xx = x.repeat(len(y))
        yy = y.view(-1, 1).repeat(1, len(x)).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def grid_anchors(self, featmap_size, stride=16, device=None):
        device = torch.device(device) if device is not None else self.base_anchors.device
        base = self._dev_cache.setdefault(str(device), self.base_anchors.to(device))
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0, feat_w, device=device) * stride
        shift_y = torch.arange(0, feat_h, device=device) * stride
        shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
        zeros = torch.zeros_like(shift_xx)
        shifts = torch.stack([shift_xx, shift_yy, zeros, zeros, zeros], dim=-1).to(base.dtype)
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 5)

    def valid_flags(self, featmap_size, valid_size, device=None):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        valid_x = torch.zeros((feat_w,), dtype=torch.bool, device=device)
        valid_y = torch.zeros((feat_h,), dtype=torch.bool, device=device)
        valid_x[:valid_w] = True
        valid_y[:valid_h] = True
        valid_xx, valid_yy = self._meshgrid(valid_x, valid_y)
        valid = valid_xx & valid_yy
        return valid[:, None].expand(valid.size(0), self.num_base_anchors).reshape(-1)


@BOXES.register_module()
class AnchorGeneratorRotatedS2ANet(_RotatedAnchorBase):
    def __init__(self, base_size, scales, ratios, angles=[0, ], scale_major=True, ctr=None):
        self._finish(base_size, scales, ratios, angles, scale_major, ctr)


@BOXES.register_module()
class AnchorGeneratorRotatedRetinaNet(_RotatedAnchorBase):
    def __init__(self, base_size, scales, ratios, angles=[0, ], octave_base_scale=None, scales_per_octave=None,
                 scale_major=True, ctr=None):
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None)), \
            "scales and octave_base_scale with scales_per_octave cannot be set at the same time"
        if scales is None:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            scales = octave_scales * octave_base_scale
        self._finish(base_size, scales, ratios, angles, scale_major, ctr)
