"""LevelPack (models/utils/level_pack.py) on the host: placement invariants, the row mask the fused conv multiplies into
its output rows, and pack / unpack as autograd Functions against plain slicing (values and gradients)."""
import pytest
import torch

from jdet_amd.models.utils import level_pack
from jdet_amd.models.utils.level_pack import LevelPack

SIZES = [[(64, 64), (32, 32), (16, 16), (8, 8)], [(32, 32), (16, 16), (8, 8)], [(8, 8), (4, 4)], [(5, 9), (3, 4), (2, 2), (1, 1)]]


@pytest.mark.parametrize("sizes", SIZES)
def test_placement_keeps_a_gap_between_levels_and_the_mask_marks_the_payload(sizes):
    p = LevelPack(sizes, torch.device("cpu"))
    occ = torch.zeros(p.height + 2, p.width + 2, dtype=torch.int32)
    for (h, w), (r0, c0) in zip(p.sizes, p.places):
        assert r0 >= 0 and c0 >= 0 and r0 + h <= p.height and c0 + w <= p.width
        # the level dilated by one position must not meet another level's dilated-by-zero payload
        occ[r0:r0 + h + 2, c0:c0 + w + 2] += 1
    inner = torch.zeros_like(occ)
    for (h, w), (r0, c0) in zip(p.sizes, p.places):
        inner[r0 + 1:r0 + 1 + h, c0 + 1:c0 + 1 + w] += 1
    assert int(inner.max()) == 1                               # payloads do not overlap
    assert int(((occ > 1) & (inner > 0)).sum()) == 0           # nobody's one-position halo touches a payload
    assert int(p.mask.sum()) == sum(h * w for h, w in sizes)
    rows = p.row_mask(3)
    assert rows.shape == (3 * p.height * p.width,) and rows.dtype == torch.float32
    assert torch.equal(rows.view(3, p.height, p.width)[1], p.mask[0, 0].float())
    assert p.row_mask(3) is rows                               # cached per batch size


@pytest.mark.parametrize("sizes", SIZES)
@pytest.mark.parametrize("channels_last", [False, True])
def test_pack_and_unpack_functions_equal_plain_slicing(sizes, channels_last, monkeypatch):
    torch.manual_seed(len(sizes))
    p = LevelPack(sizes, torch.device("cpu"))
    base = [torch.randn(2, 6, h, w) for h, w in sizes]
    if channels_last:
        base = [b.contiguous(memory_format=torch.channels_last) for b in base]
    weight = torch.randn(2, 6, p.height, p.width)
    results = []
    for functions in (True, False):
        monkeypatch.setattr(level_pack, "PACK_FUNCTIONS", functions)
        xs = [b.clone().requires_grad_(True) for b in base]
        y = p.pack(xs)
        assert y.shape == (2, 6, p.height, p.width)
        assert y.is_contiguous(memory_format=torch.channels_last) == channels_last or not channels_last
        outs = p.unpack(y * weight)
        loss = sum((o * o).sum() * (i + 1) for i, o in enumerate(outs))
        loss.backward()
        results.append((y.detach(), [o.detach() for o in outs], [x.grad for x in xs]))
    (y1, o1, g1), (y0, o0, g0) = results
    assert torch.equal(y1, y0)
    assert torch.equal(y1 * (1 - p.mask.float()), torch.zeros_like(y1))        # gaps are zero
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


def test_unused_level_gradients_are_zero():
    """a level whose unpacked output takes no part in the loss gets zeros in the canvas gradient"""
    sizes = [(8, 8), (4, 4), (2, 2)]
    p = LevelPack(sizes, torch.device("cpu"))
    xs = [torch.randn(1, 2, h, w, requires_grad=True) for h, w in sizes]
    outs = p.unpack(p.pack(xs) * 3.0)
    outs[1].sum().backward()
    assert torch.equal(xs[0].grad, torch.zeros_like(xs[0])) and torch.equal(xs[2].grad, torch.zeros_like(xs[2]))
    assert torch.equal(xs[1].grad, torch.full_like(xs[1], 3.0))
