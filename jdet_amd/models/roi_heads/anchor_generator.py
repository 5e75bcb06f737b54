"""Single-level horizontal anchors of AnchorHead / FasterrcnnHead (python/jdet/models/roi_heads/anchor_generator.py:L553-631):
corner boxes of side base * scale and aspect ratio `ratio` around the pixel-centre (base - 1) / 2, with the legacy "- 1"
extent and ROUNDED corners; lattice and validity mask are the closed forms of models/boxes/anchor_generator.py."""
import torch

from jdet_amd.models.boxes.anchor_generator import _aspect_sides, _inside, _lattice


class AnchorGenerator:
    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size, self.scale_major, self.ctr = base_size, scale_major, ctr
        self.scales = torch.as_tensor(scales, dtype=torch.float32)
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32)
        self.base_anchors = self.gen_base_anchors()
        self._kept = {}      # lattices / masks per (size, stride, device): tiles of one shape need them once

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        side = self.base_size
        cx, cy = (0.5 * (side - 1),) * 2 if self.ctr is None else self.ctr
        ws, hs = _aspect_sides(side, self.ratios, self.scales, self.scale_major)
        half_w, half_h = 0.5 * (ws - 1), 0.5 * (hs - 1)      # inclusive pixel extents
        return torch.stack([cx - half_w, cy - half_h, cx + half_w, cy + half_h], dim=-1).round().float()

    def grid_anchors(self, featmap_size, stride=16, device=None):
        key = (tuple(featmap_size), stride, str(device))
        hit = self._kept.get(key)
        if hit is None:
            hit = self._kept[key] = _lattice(self.base_anchors.to(device), featmap_size[0], featmap_size[1], stride, stride,
                                             (0, 2), (1, 3))
        return hit

    def valid_flags(self, featmap_size, valid_size, device=None):
        key = ("valid", tuple(featmap_size), tuple(valid_size), str(device))
        hit = self._kept.get(key)
        if hit is None:
            hit = self._kept[key] = _inside(featmap_size[0], featmap_size[1], valid_size[0], valid_size[1],
                                            self.num_base_anchors, device)
        return hit
