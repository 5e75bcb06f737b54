"""Running weight-shared head convolutions once for several small pyramid levels.

The dense heads apply the SAME conv towers to every FPN level.  On the small levels (32x32 and below at a 1024 tile) a
3x3 convolution over 256 channels is latency bound -- a 2304-deep reduction for a handful of output tiles, ~48 us per
launch whatever the map size -- and every level adds its own data-gradient, weight-gradient and gradient-accumulation
launches.  `LevelPack` places such levels in one tensor; a tower then costs one launch per layer for all of them.
"""
import os

import torch

PACK_FUNCTIONS = os.environ.get("JDET_PACK_FUNCTIONS", "1") == "1"      # A/B switch (see _Pack / _Unpack)
FUSED_PACK = os.environ.get("JDET_PACK_FUSED_FILL", "1") == "1"          # canvas + windows in one kernel (A/B switch)


class LevelPack:
    """Several (h, w) maps placed in one (N, C, Hp, Wp) tensor with at least one empty row / column between any two of
    them.  The gaps are kept at zero (`mask`), so a 3x3 / padding 1 convolution of the packed tensor sees at every
    level border exactly the zeros its own zero padding would supply.
    Placement (round 4): the largest level at the top left, the others in a column to its right -- the next one on
    top, the rest on shelves below it, left to right -- e.g. 64^2, 32^2, 16^2, 8^2 -> 64 x 97 = 6208 positions for 5456
    of payload (14 % padding; stacked along H as in rounds 2-3 the same levels take 123 x 64 = 7872: 44 %).  Levels of
    one width still stack along H when that is tighter."""

    def __init__(self, sizes, device):
        self.sizes = [tuple(s) for s in sizes]
        self.places = self._place(self.sizes)
        self.height = max(r + h for (r, _), (h, _) in zip(self.places, self.sizes))
        self.width = max(c + w for (_, c), (_, w) in zip(self.places, self.sizes))
        mask = torch.zeros((1, 1, self.height, self.width), dtype=torch.bool, device=device)
        for (h, w), (r0, c0) in zip(self.sizes, self.places):
            mask[:, :, r0:r0 + h, c0:c0 + w] = True
        self.mask = mask
        self._rows = {}

    def row_mask(self, batch):
        """the mask per position of a batch, flat fp32 (batch * height * width): what the fused conv multiplies into its
        finished rows (ops/conv_igemm: rowmask)"""
        rows = self._rows.get(batch)
        if rows is None:
            while len(self._rows) >= 4:
                self._rows.pop(next(iter(self._rows)))
            rows = self._rows[batch] = self.mask.to(torch.float32).expand(batch, 1, self.height, self.width).reshape(-1).contiguous()
        return rows

    @staticmethod
    def _stack(sizes):
        places, r = [], 0
        for h, _ in sizes:
            places.append((r, 0))
            r += h + 1
        return places

    @staticmethod
    def _place(sizes):
        stacked = LevelPack._stack(sizes)
        if len(sizes) < 3:
            return stacked
        order = sorted(range(len(sizes)), key=lambda i: -sizes[i][0] * sizes[i][1])
        first = order[0]
        h0, w0 = sizes[first]
        places = [None] * len(sizes)
        places[first] = (0, 0)
        c0 = w0 + 1
        # the second level on top of the side column; the rest on shelves below it
        second = order[1]
        places[second] = (0, c0)
        col_w = sizes[second][1]
        r, c, shelf_h = sizes[second][0] + 1, c0, 0
        for i in order[2:]:
            h, w = sizes[i]
            if c + w > c0 + col_w and c > c0:        # shelf full: next shelf
                r, c, shelf_h = r + shelf_h + 1, c0, 0
            places[i] = (r, c)
            c += w + 1
            shelf_h = max(shelf_h, h)
            col_w = max(col_w, c - 1 - c0)
        area = lambda pl: (max(r_ + h_ for (r_, _), (h_, _) in zip(pl, sizes)) *
                           max(c_ + w_ for (_, c_), (_, w_) in zip(pl, sizes)))
        return places if area(places) < area(stacked) else stacked

    _cache = {}
    CACHE_MAX = 32          # distinct (level sizes, device) layouts kept (multi-scale training cycles through a few)

    @classmethod
    def cached(cls, sizes, device):
        key = (tuple(tuple(s) for s in sizes), str(device))
        hit = cls._cache.pop(key, None)
        if hit is None:
            hit = cls(sizes, device)
            while len(cls._cache) >= cls.CACHE_MAX:          # least recently used first (dict order = recency)
                cls._cache.pop(next(iter(cls._cache)))
        cls._cache[key] = hit
        return hit

    def _canvas(self, like, channels):
        cl = like.dim() == 4 and like.is_contiguous(memory_format=torch.channels_last)
        return self._canvas_like(like.shape[0], channels, like.dtype, like.device, cl)

    def _canvas_like(self, batch, channels, dtype, device, channels_last):
        canvas = torch.empty((batch, channels, self.height, self.width), dtype=dtype, device=device,
                             memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        if canvas.is_cuda and canvas.element_size() == 4:
            from jdet_amd import _lib as L
            return L.zero_(canvas)      # a plain kernel: the framework zero-fills a tensor of this size with a memset (node)
        return canvas.zero_()

    def fill(self, xs, batch, channels, dtype, device):
        """the canvas with level i's window = xs[i] (None: zeros) and zero gaps.  fp32 channels-last device maps with
        C % 4 == 0: ONE kernel writes every word of the canvas once (csrc/level_pack.hip); otherwise a zero canvas + one
        window copy per level."""
        import ctypes
        canvas = torch.empty((batch, channels, self.height, self.width), dtype=dtype, device=device,
                             memory_format=torch.channels_last)
        fused = FUSED_PACK and canvas.is_cuda and dtype == torch.float32 and channels % 4 == 0 and len(self.sizes) <= 8
        if fused:
            from jdet_amd import _lib as L
            keep = []
            for x in xs:
                if x is not None and not (x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
                    x = x.to(torch.float32).contiguous(memory_format=torch.channels_last)
                keep.append(x)
            n = len(self.sizes)
            ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() if x is not None else None for x in keep])
            hw = (ctypes.c_int32 * (2 * n))(*[v for s in self.sizes for v in s])
            place = (ctypes.c_int32 * (2 * n))(*[v for pl in self.places for v in pl])
            L.check(L.lib().jdet_level_pack_nhwc(ptrs, hw, place, n, batch, channels, self.height, self.width,
                                                 canvas.data_ptr(), L.stream_ptr(canvas)), "jdet_level_pack_nhwc")
            return canvas
        canvas = self._canvas_like(batch, channels, dtype, device, True)
        for x, dst in zip(xs, self._slices(canvas)):
            if x is not None:
                dst.copy_(x)
        return canvas

    def _slices(self, y):
        return [y[:, :, r0:r0 + h, c0:c0 + w] for (h, w), (r0, c0) in zip(self.sizes, self.places)]

    def pack(self, xs):
        if PACK_FUNCTIONS and torch.is_grad_enabled() and any(x.requires_grad for x in xs):
            return _Pack.apply(self, *xs)
        out = self._canvas(xs[0], xs[0].shape[1])
        for x, (h, w), (r0, c0) in zip(xs, self.sizes, self.places):
            out[:, :, r0:r0 + h, c0:c0 + w] = x
        return out

    def unpack(self, y):
        if PACK_FUNCTIONS and torch.is_grad_enabled() and y.requires_grad:
            return list(_Unpack.apply(self, y))
        return self._slices(y)


class _Pack(torch.autograd.Function):
    """canvas of zeros + one copy per level; backward: the levels' windows of the canvas gradient (views).  (Written
    as slice assignments the framework records a CopySlices node per level, each with its own backward launches.)"""

    @staticmethod
    def forward(ctx, pack, *xs):
        ctx.pack = pack
        x0 = xs[0]
        if x0.dim() == 4 and x0.is_contiguous(memory_format=torch.channels_last):
            return pack.fill(xs, x0.shape[0], x0.shape[1], x0.dtype, x0.device)
        out = pack._canvas(x0, x0.shape[1])
        for x, dst in zip(xs, pack._slices(out)):
            dst.copy_(x)
        return out

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(ctx.pack._slices(g))


class _Unpack(torch.autograd.Function):
    """the levels' windows of a packed tensor (views); backward: ONE zero canvas + one copy per level -- the
    framework's slice backward builds a full zero canvas PER level and then adds the canvases up"""

    @staticmethod
    def forward(ctx, pack, y):
        ctx.pack = pack
        # only what the backward's canvas needs (keeping `y` itself would hold every head output's canvas until then)
        ctx.like = (y.shape, y.dtype, y.device, y.is_contiguous(memory_format=torch.channels_last))
        return tuple(pack._slices(y))

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device, cl = ctx.like
        if cl:
            return None, ctx.pack.fill(grads, shape[0], shape[1], dtype, device)
        g = ctx.pack._canvas_like(shape[0], shape[1], dtype, device, cl)
        for gl, dst in zip(grads, ctx.pack._slices(g)):
            if gl is not None:
                dst.copy_(gl)
        return None, g


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
