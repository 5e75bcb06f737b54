"""Running weight-shared head convolutions once for several small pyramid levels.

The dense heads apply the SAME conv towers to every FPN level.  On the small levels (32x32 and below at a 1024 tile) a
3x3 convolution over 256 channels is latency bound -- a 2304-deep reduction for a handful of output tiles, ~48 us per
launch whatever the map size -- and every level adds its own data-gradient, weight-gradient and gradient-accumulation
launches.  `LevelPack` stacks such levels into one tensor; a tower then costs one launch per layer for all of them.
"""
import torch
import torch.nn.functional as F


class LevelPack:
    """Several (h, w) maps stacked along H in one (N, C, Hp, Wp) tensor: level l occupies rows r_l .. r_l + h_l and
    columns 0 .. w_l, one empty row between levels, narrower levels padded on the right.  The gaps are kept at zero
    (`mask`), so a 3x3 / padding 1 convolution of the packed tensor sees at every level border exactly the zeros
    its own zero padding would supply."""

    def __init__(self, sizes, device):
        self.sizes = [tuple(s) for s in sizes]
        self.width = max(w for _, w in self.sizes)
        self.rows, r = [], 0
        for h, _ in self.sizes:
            self.rows.append(r)
            r += h + 1
        self.height = r - 1
        mask = torch.zeros((1, 1, self.height, self.width), dtype=torch.bool, device=device)
        for (h, w), r0 in zip(self.sizes, self.rows):
            mask[:, :, r0:r0 + h, :w] = True
        self.mask = mask

    _cache = {}

    @classmethod
    def cached(cls, sizes, device):
        key = (tuple(tuple(s) for s in sizes), str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(sizes, device)
        return cls._cache[key]

    def pack(self, xs):
        parts = []
        for x, (h, w), r0 in zip(xs, self.sizes, self.rows):
            parts.append(F.pad(x, (0, self.width - w, 0, 1 if r0 + h < self.height else 0)))
        return torch.cat(parts, dim=2)

    def unpack(self, y):
        return [y[:, :, r0:r0 + h, :w] for (h, w), r0 in zip(self.sizes, self.rows)]


def run_levels(feats, fn, max_positions=1024):
    """fn(x, mask) -> tuple of tensors with x's spatial size (mask: None, or the (1,1,H,W) 0/1 tensor every 3x3 layer
    of fn must multiply its output with).  Levels of at most `max_positions` positions (device tensors, at least two
    of them) go through fn once as a LevelPack; the others one by one.  Returns a list (per level) of fn's tuples."""
    small = [i for i, f in enumerate(feats) if f.is_cuda and f.shape[-2] * f.shape[-1] <= max_positions]
    outs = [None] * len(feats)
    if len(small) >= 2:
        pack = LevelPack.cached([tuple(feats[i].shape[-2:]) for i in small], feats[small[0]].device)
        packed = fn(pack.pack([feats[i] for i in small]), pack.mask.to(feats[small[0]].dtype))
        per_output = [pack.unpack(t) if t is not None else [None] * len(small) for t in packed]
        for k, i in enumerate(small):
            outs[i] = tuple(o[k] for o in per_output)
    for i, f in enumerate(feats):
        if outs[i] is None:
            outs[i] = tuple(fn(f, None))
    return outs
