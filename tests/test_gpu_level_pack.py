"""GPU: the one-launch LevelPack fill of round 6 (csrc/level_pack.hip; pack forward, unpack backward) against the framework
composition it replaces (zero canvas + one window copy per level).  Copies: results EQUAL."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [[(64, 64), (32, 32), (16, 16), (8, 8)], [(32, 32), (16, 16), (8, 8)], [(8, 8), (4, 4)], [(5, 9), (3, 4), (2, 2), (1, 1)]]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("sizes", SIZES)
@pytest.mark.parametrize("C", [256, 32, 4])
def test_fused_fill_equals_zero_canvas_and_window_copies(dev, sizes, C, monkeypatch):
    from jdet_amd.models.utils import level_pack
    from jdet_amd.models.utils.level_pack import LevelPack
    torch.manual_seed(C + len(sizes))
    p = LevelPack(sizes, dev)
    base = [torch.randn(2, C, h, w, device=dev).contiguous(memory_format=torch.channels_last) for h, w in sizes]
    weight = torch.randn(2, C, p.height, p.width, device=dev).contiguous(memory_format=torch.channels_last)
    results = []
    for fused in (True, False):
        monkeypatch.setattr(level_pack, "FUSED_PACK", fused)
        xs = [b.clone().requires_grad_(True) for b in base]
        y = p.pack(xs)
        assert y.is_contiguous(memory_format=torch.channels_last)
        outs = p.unpack(y * weight)
        # level 1 takes no part in the loss: its window of the canvas gradient must be zeros (a NULL source)
        loss = sum((o * o).sum() * (i + 1) for i, o in enumerate(outs) if i != 1)
        loss.backward()
        results.append((y.detach(), [x.grad for x in xs]))
    (y1, g1), (y0, g0) = results
    assert torch.equal(y1, y0)
    assert torch.equal(y1 * (~p.mask).float(), torch.zeros_like(y1))
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
    assert torch.equal(g1[1], torch.zeros_like(g1[1]))


def test_fused_fill_refusals(dev):
    import ctypes
    from jdet_amd import _lib as L
    lib = L.lib()
    x = torch.zeros(1, 8, 8, 8, device=dev)
    ptrs = (ctypes.c_void_p * 1)(x.data_ptr())
    hw, place = (ctypes.c_int32 * 2)(8, 8), (ctypes.c_int32 * 2)(0, 0)
    out = torch.zeros(1, 9, 9, 8, device=dev)
    assert lib.jdet_level_pack_nhwc(ptrs, hw, place, 1, 1, 8, 9, 9, out.data_ptr(), None) == 0
    assert lib.jdet_level_pack_nhwc(ptrs, hw, place, 1, 1, 6, 9, 9, out.data_ptr(), None) == -2      # C % 4: unsupported
    place_bad = (ctypes.c_int32 * 2)(2, 0)                                       # window past the canvas
    assert lib.jdet_level_pack_nhwc(ptrs, hw, place_bad, 1, 1, 8, 9, 9, out.data_ptr(), None) == -1
    assert lib.jdet_level_pack_nhwc(ptrs, hw, place, 9, 1, 8, 9, 9, out.data_ptr(), None) == -1
