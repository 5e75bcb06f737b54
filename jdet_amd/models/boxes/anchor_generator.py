"""Anchor lattices of the dense heads, behind the reference's class names and call signatures
(python/jdet/models/boxes/anchor_generator.py: `AnchorGeneratorRotatedRetinaNet` L7-110, `AnchorGeneratorRotatedS2ANet`
L112-196, `AnchorGenerator` L198-597).

Every generator is the same two closed forms, written once here:
  * a lattice:  anchor[(y * W + x) * A + a] = base[a] + (x * stride_x, y * stride_y) added to the base's position columns
    -- locations row-major, the A base anchors fastest -- built as one broadcast over an (H, W, A, D) view;
  * a validity mask: flag[(y * W + x) * A + a] = (y < valid_h) and (x < valid_w).
The base anchors keep the reference's arithmetic (operation order included: the S2ANet / RetinaNet targets are compared
bit for bit against the restatement in oracle/box_oracle.py): rotated ones are (xc, yc, w, h, angle) with the centre at
(base - 1) / 2, ratio-major, then scale, then angle; horizontal ones are corner boxes around center_offset * base.
Lattices are cached per (size, stride, device): tiles of one shape need them once."""
import numpy as np
import torch

from jdet_amd.utils.registry import BOXES


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _octave_scales(octave_base_scale, scales_per_octave):
    return np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)]) * octave_base_scale


def _scales_from(scales, octave_base_scale, scales_per_octave):
    octave = octave_base_scale is not None and scales_per_octave is not None
    assert octave ^ (scales is not None), \
        "give either `scales` or `octave_base_scale` with `scales_per_octave`, not both"
    return torch.as_tensor(_octave_scales(octave_base_scale, scales_per_octave) if scales is None else scales,
                           dtype=torch.float32)


def _aspect_sides(side, ratios, scales, scale_major=True):
    """(widths, heights) of side * scale boxes of aspect ratio h / w = ratio, flattened ratio-major (scale_major) or
    scale-major; the products are taken in the reference's order: (side * ratio term) * scale"""
    rh = torch.sqrt(ratios)
    rw = 1 / rh
    if scale_major:
        return (side * rw[:, None] * scales[None, :]).reshape(-1), (side * rh[:, None] * scales[None, :]).reshape(-1)
    return (side * scales[:, None] * rw[None, :]).reshape(-1), (side * scales[:, None] * rh[None, :]).reshape(-1)


def _lattice(base, feat_h, feat_w, stride_x, stride_y, x_cols, y_cols):
    """base (A, D) on its device -> (H * W * A, D): base shifted by (x * stride_x, y * stride_y) in the given columns"""
    dev, A, D = base.device, base.shape[0], base.shape[1]
    shift = torch.zeros((feat_h, feat_w, 1, D), dtype=base.dtype, device=dev)
    xs = (torch.arange(feat_w, device=dev) * stride_x).to(base.dtype)
    ys = (torch.arange(feat_h, device=dev) * stride_y).to(base.dtype)
    for c in x_cols:
        shift[:, :, 0, c] = xs[None, :]
    for c in y_cols:
        shift[:, :, 0, c] = ys[:, None]
    return (base.view(1, 1, A, D) + shift).reshape(-1, D)


def _inside(feat_h, feat_w, valid_h, valid_w, num_base, device):
    assert valid_h <= feat_h and valid_w <= feat_w
    rows = torch.arange(feat_h, device=device) < valid_h
    cols = torch.arange(feat_w, device=device) < valid_w
    return (rows[:, None] & cols[None, :]).reshape(-1, 1).expand(feat_h * feat_w, num_base).reshape(-1)


class _RotatedLattice:
    """(xc, yc, w, h, angle) anchors of one level: S2ANet (one square per location) and RetinaNet-OBB (octave scales x
    ratios [x angles]) differ only in how their constructor names the scales"""

    def _setup(self, base_size, scales, ratios, angles, scale_major, ctr):
        assert scale_major, "AnchorGeneratorRotated only support scale-major anchors!"
        self.base_size, self.scale_major, self.ctr = base_size, scale_major, ctr
        self.scales = torch.as_tensor(scales, dtype=torch.float32)
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32)
        self.angles = torch.as_tensor(angles, dtype=torch.float32)
        self.base_anchors = self.gen_base_anchors()
        self._on_device = {}

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        side = self.base_size
        cx, cy = (0.5 * (side - 1), 0.5 * (side - 1)) if self.ctr is None else self.ctr
        ws, hs = _aspect_sides(side, self.ratios, self.scales)
        n_ang = len(self.angles)
        one = torch.ones_like(self.angles)
        ws = (ws[:, None] * one[None, :]).reshape(-1)          # ratio-major, then scale, then angle
        hs = (hs[:, None] * one[None, :]).reshape(-1)
        ang = self.angles.repeat(ws.numel() // n_ang)
        return torch.stack([cx + torch.zeros_like(ws), cy + torch.zeros_like(ws), ws, hs, ang], dim=-1)

    def grid_anchors(self, featmap_size, stride=16, device=None):
        device = torch.device(device) if device is not None else self.base_anchors.device
        base = self._on_device.get(str(device))
        if base is None:
            base = self._on_device[str(device)] = self.base_anchors.to(device)
        return _lattice(base, featmap_size[0], featmap_size[1], stride, stride, (0,), (1,))

    def valid_flags(self, featmap_size, valid_size, device=None):
        return _inside(featmap_size[0], featmap_size[1], valid_size[0], valid_size[1], self.num_base_anchors, device)


@BOXES.register_module()
class AnchorGeneratorRotatedS2ANet(_RotatedLattice):
    def __init__(self, base_size, scales, ratios, angles=[0, ], scale_major=True, ctr=None):
        self._setup(base_size, scales, ratios, angles, scale_major, ctr)


@BOXES.register_module()
class AnchorGeneratorRotatedRetinaNet(_RotatedLattice):
    def __init__(self, base_size, scales, ratios, angles=[0, ], octave_base_scale=None, scales_per_octave=None,
                 scale_major=True, ctr=None):
        self._setup(base_size, _scales_from(scales, octave_base_scale, scales_per_octave), ratios, angles, scale_major,
                    ctr)


@BOXES.register_module()
class AnchorGenerator:
    """Multi-level horizontal anchors <x1, y1, x2, y2> (anchor_generator.py:L198-597): per level a base set around
    `center_offset * base_size` (or an explicit centre), one lattice per feature map."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if not 0 <= center_offset <= 1:
            raise ValueError(f"center_offset should be in range [0, 1], {center_offset} is given.")
        assert center_offset == 0 or centers is None, "give `centers` or a non-zero `center_offset`, not both"
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides), "one base size per stride"
        assert centers is None or len(centers) == len(self.strides), "one centre per stride"
        self.scales = _scales_from(scales, octave_base_scale, scales_per_octave)
        self.octave_base_scale, self.scales_per_octave = octave_base_scale, scales_per_octave
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32)
        self.scale_major, self.centers, self.center_offset = scale_major, centers, center_offset
        self.base_anchors = self.gen_base_anchors()
        self._lattices = {}

    @property
    def num_levels(self):
        return len(self.strides)

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    num_base_priors = num_base_anchors

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        cx, cy = (self.center_offset * base_size,) * 2 if center is None else center
        ws, hs = _aspect_sides(base_size, ratios, scales, self.scale_major)
        return torch.stack([cx - 0.5 * ws, cy - 0.5 * hs, cx + 0.5 * ws, cy + 0.5 * hs], dim=-1)

    def gen_base_anchors(self):
        return [self.gen_single_level_base_anchors(b, self.scales, self.ratios,
                                                   None if self.centers is None else self.centers[lvl])
                for lvl, b in enumerate(self.base_sizes)]

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device=None):
        key = (id(base_anchors), tuple(featmap_size), tuple(stride), str(device))
        hit = self._lattices.get(key)
        if hit is None:
            base = base_anchors if device is None else base_anchors.to(device)
            hit = self._lattices[key] = _lattice(base, featmap_size[0], featmap_size[1], stride[0], stride[1], (0, 2),
                                                 (1, 3))
        return hit

    def grid_anchors(self, featmap_sizes, device=None):
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_grid_anchors(self.base_anchors[lvl], featmap_sizes[lvl], self.strides[lvl], device)
                for lvl in range(self.num_levels)]

    grid_priors = grid_anchors

    def single_level_valid_flags(self, featmap_size, valid_size, num_base_anchors, device=None):
        return _inside(featmap_size[0], featmap_size[1], valid_size[0], valid_size[1], num_base_anchors, device)

    def valid_flags(self, featmap_sizes, pad_shape, device=None):
        assert self.num_levels == len(featmap_sizes)
        img_h, img_w = pad_shape[:2]
        flags = []
        for lvl, (feat_h, feat_w) in enumerate(featmap_sizes):
            sx, sy = self.strides[lvl]
            covered = (min(int(np.ceil(img_h / sy)), feat_h), min(int(np.ceil(img_w / sx)), feat_w))
            flags.append(self.single_level_valid_flags((feat_h, feat_w), covered, self.num_base_anchors[lvl], device))
        return flags
