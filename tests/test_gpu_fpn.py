"""GPU: the fused top-down step of the FPN (csrc/upsample_add.hip, ops/upsample_add.py) against the two framework ops it
replaces (fpn.py:L160-171: nearest interpolate + add [+ divide]) -- values and both gradients, exact 2x and odd sizes --
and the FPN built on it against the same module with the fused step disabled."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C,H,W,Ht,Wt,div", [(2, 256, 64, 64, 32, 32, 1.0), (1, 64, 25, 38, 13, 19, 1.0),
                                               (2, 32, 17, 9, 5, 4, 2.0), (1, 8, 7, 7, 7, 7, 1.0),
                                               (1, 16, 12, 20, 4, 5, 1.0)])
def test_upsample_add_equals_interpolate_plus_add(dev, N, C, H, W, Ht, Wt, div):
    from jdet_amd.ops import upsample_add as UA
    g = torch.Generator(device="cpu").manual_seed(H * W + C)
    lat = torch.randn((N, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    top = torch.randn((N, C, Ht, Wt), generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn((N, C, H, W), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    assert UA.fusable(lat, top)
    y = UA.upsample_add(lat, top, div)
    y.backward(go)
    got = (y.detach().clone(), lat.grad.clone(), top.grad.clone())
    lat.grad = top.grad = None
    ref = (lat + F.interpolate(top, size=(H, W), mode="nearest")) / div
    ref.backward(go)
    assert torch.equal(got[0], ref.detach())                       # one add (and one divide) per element: same bits
    assert torch.equal(got[1], lat.grad)
    torch.testing.assert_close(got[2], top.grad, rtol=1e-6, atol=1e-6)      # a sum of <= a few terms, another order


@pytest.mark.parametrize("extra", [False, "on_input", "on_output"])
def test_fpn_with_the_fused_step_equals_the_framework_ops(dev, extra, monkeypatch):
    from jdet_amd.models.necks.fpn import FPN
    from jdet_amd.ops import upsample_add as UA
    torch.manual_seed(3)
    m = FPN([16, 32, 64, 128], 32, 5, start_level=1 if extra else 0, add_extra_convs=extra).to(dev).to(
        memory_format=torch.channels_last)
    xs = [torch.randn((2, c, s, s), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for c, s in ((16, 64), (32, 32), (64, 16), (128, 8))]

    def run():
        outs = m(xs)
        sum(o.square().sum() for o in outs).backward()
        res = [o.detach().clone() for o in outs], [x.grad.clone() for x in xs if x.grad is not None], \
            [p.grad.clone() for p in m.parameters()]
        for x in xs:
            x.grad = None
        m.zero_grad(set_to_none=True)
        return res
    a = run()
    monkeypatch.setattr(UA, "fusable", lambda lat, top: False)
    b = run()
    assert len(a[0]) == 5
    for u, v in zip(a[0], b[0]):
        torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-5)
    for grp in (1, 2):
        for u, v in zip(a[grp], b[grp]):
            torch.testing.assert_close(u, v, rtol=2e-4, atol=2e-4 * float(v.abs().max()))
