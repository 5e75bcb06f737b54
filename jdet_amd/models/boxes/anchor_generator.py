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


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@BOXES.register_module()
class AnchorGenerator:
    """Standard multi-level horizontal anchor generator (anchor_generator.py:L198-597): base anchors
    <x1,y1,x2,y2> around `center_offset * base_size`, ratio-major then scale when `scale_major`."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None, f"center cannot be set when center_offset!=0, {centers} is given."
        if not (0 <= center_offset <= 1):
            raise ValueError(f"center_offset should be in range [0, 1], {center_offset} is given.")
        if centers is not None:
            assert len(centers) == len(strides), \
                f"The number of strides should be the same as centers, got {strides} and {centers}"
        self.strides = [_pair(stride) for stride in strides]
        self.base_sizes = [min(stride) for stride in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides), \
            f"The number of strides should be the same as base sizes, got {self.strides} and {self.base_sizes}"
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None)), \
            "scales and octave_base_scale with scales_per_octave cannot be set at the same time"
        if scales is not None:
            self.scales = torch.as_tensor(scales, dtype=torch.float32)
        else:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.as_tensor(octave_scales * octave_base_scale, dtype=torch.float32)
        self.octave_base_scale = octave_base_scale
        self.scales_per_octave = scales_per_octave
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32)
        self.scale_major = scale_major
        self.centers = centers
        self.center_offset = center_offset
        self.base_anchors = self.gen_base_anchors()
        self._dev_cache = {}

    @property
    def num_base_anchors(self):
        return [base_anchors.size(0) for base_anchors in self.base_anchors]

    num_base_priors = num_base_anchors

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        return [self.gen_single_level_base_anchors(base_size, self.scales, self.ratios,
                                                   self.centers[i] if self.centers is not None else None)
                for i, base_size in enumerate(self.base_sizes)]

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w = h = base_size
        if center is None:
            x_center, y_center = self.center_offset * w, self.center_offset * h
        else:
            x_center, y_center = center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_center - 0.5 * ws, y_center - 0.5 * hs, x_center + 0.5 * ws, y_center + 0.5 * hs], dim=-1)

    @staticmethod
    def _meshgrid(x, y, row_major=True):
        xx = x.repeat(y.shape[0])
        yy = y.view(-1, 1).repeat(1, x.shape[0]).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def grid_anchors(self, featmap_sizes, device=None):
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_grid_anchors(self.base_anchors[i], featmap_sizes[i], self.strides[i], device)
                for i in range(self.num_levels)]

    grid_priors = grid_anchors

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device=None):
        key = (id(base_anchors), tuple(featmap_size), tuple(stride), str(device))
        if key in self._dev_cache:
            return self._dev_cache[key]
        base = base_anchors.to(device) if device is not None else base_anchors
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0, feat_w, device=base.device) * stride[0]
        shift_y = torch.arange(0, feat_h, device=base.device) * stride[1]
        shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
        shifts = torch.stack([shift_xx, shift_yy, shift_xx, shift_yy], dim=-1).to(base.dtype)
        out = (base[None, :, :] + shifts[:, None, :]).view(-1, 4)
        self._dev_cache[key] = out
        return out

    def valid_flags(self, featmap_sizes, pad_shape, device=None):
        assert self.num_levels == len(featmap_sizes)
        multi_level_flags = []
        for i in range(self.num_levels):
            anchor_stride = self.strides[i]
            feat_h, feat_w = featmap_sizes[i]
            h, w = pad_shape[:2]
            valid_feat_h = min(int(np.ceil(h / anchor_stride[1])), feat_h)
            valid_feat_w = min(int(np.ceil(w / anchor_stride[0])), feat_w)
            multi_level_flags.append(self.single_level_valid_flags((feat_h, feat_w), (valid_feat_h, valid_feat_w),
                                                                   self.num_base_anchors[i], device))
        return multi_level_flags

    def single_level_valid_flags(self, featmap_size, valid_size, num_base_anchors, device=None):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        valid_x = torch.zeros(feat_w, dtype=torch.bool, device=device)
        valid_y = torch.zeros(feat_h, dtype=torch.bool, device=device)
        valid_x[:valid_w] = True
        valid_y[:valid_h] = True
        valid_xx, valid_yy = self._meshgrid(valid_x, valid_y)
        valid = valid_xx & valid_yy
        return valid[:, None].expand(valid.size(0), num_base_anchors).reshape(-1)
