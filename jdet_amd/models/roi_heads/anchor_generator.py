"""Single-level horizontal anchor generator used by AnchorHead / FasterrcnnHead.  Mirrors
python/jdet/models/roi_heads/anchor_generator.py:L553-631 (base anchors rounded, ctr = (base-1)/2,
ratio-major then scale, shifts row-major with the A base anchors fastest)."""
import torch


class AnchorGenerator:
    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size = base_size
        self.scales = torch.as_tensor(scales, dtype=torch.float32)
        self.ratios = torch.as_tensor(ratios, dtype=torch.float32)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()
        self._cache = {}

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
        if self.scale_major:
            ws = (w * w_ratios[:, None] * self.scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * self.scales[None, :]).view(-1)
        else:
            ws = (w * self.scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * self.scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1),
                            y_ctr + 0.5 * (hs - 1)], dim=-1).round().float()

    @staticmethod
    def _meshgrid(x, y, row_major=True):
        xx = x.repeat(len(y))
        yy = y.view(-1, 1).repeat(1, len(x)).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def grid_anchors(self, featmap_size, stride=16, device=None):
        key = (tuple(featmap_size), stride, str(device))
        if key not in self._cache:   # tiles have one shape: computed once per (level, device)
            base_anchors = self.base_anchors.to(device)
            feat_h, feat_w = featmap_size
            shift_x = torch.arange(0, feat_w, device=device) * stride
            shift_y = torch.arange(0, feat_h, device=device) * stride
            shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
            shifts = torch.stack([shift_xx, shift_yy, shift_xx, shift_yy], dim=-1).to(base_anchors.dtype)
            self._cache[key] = (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)
        return self._cache[key]

    def valid_flags(self, featmap_size, valid_size, device=None):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        key = ("valid", tuple(featmap_size), tuple(valid_size), str(device))
        if key in self._cache:
            return self._cache[key]
        valid_x = torch.zeros(feat_w, dtype=torch.bool, device=device)
        valid_y = torch.zeros(feat_h, dtype=torch.bool, device=device)
        valid_x[:valid_w] = True
        valid_y[:valid_h] = True
        valid_xx, valid_yy = self._meshgrid(valid_x, valid_y)
        valid = valid_xx & valid_yy
        self._cache[key] = valid[:, None].expand(valid.shape[0], self.num_base_anchors).reshape(-1)
        return self._cache[key]
