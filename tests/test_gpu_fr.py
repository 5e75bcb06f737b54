"""GPU: feature refinement (csrc/feature_refine.hip, ops/fr.py) against the numpy restatement (oracle/fr_oracle.py;
parity unpinned by reference execution -- CUDA-only source) and the closed form on an affine map."""
import numpy as np
import pytest
import torch

from oracle import fr_oracle as FO

pytestmark = pytest.mark.gpu


def _boxes(rng, N, H, W, stride):
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    b = np.stack([ys * stride + rng.normal(0, 2.0 * stride, (N, H, W)), xs * stride + rng.normal(0, 2.0 * stride, (N, H, W)),
                  np.exp(rng.normal(np.log(4 * stride), 0.6, (N, H, W))),
                  np.exp(rng.normal(np.log(4 * stride), 0.6, (N, H, W))), rng.uniform(-1.6, 1.6, (N, H, W))], -1)
    b[0, 0, 0, :2] = -1000.0                    # samples outside the map (dropped) ...
    b[0, 0, 1, :2] = (H + 0.5) * stride         # ... and in the clamped border band
    return b.astype(np.float32)


@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("C,H,W", [(64, 24, 20), (260, 9, 13)])
def test_feature_refine_vs_oracle(dev, points, C, H, W):
    from jdet_amd.ops.fr import FR
    rng = np.random.default_rng(3 * points + C)
    N, stride = 2, 8.0
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    boxes = _boxes(rng, N, H, W, stride)
    grad = rng.standard_normal((N, C, H, W)).astype(np.float32)
    for channels_last in (False, True):
        x = torch.from_numpy(feat).to(dev)
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        y = FR(1.0 / stride, points)(x, torch.from_numpy(boxes).to(dev))
        y.backward(torch.from_numpy(grad).to(dev))
        ref = FO.feature_refine_forward(feat, boxes, 1.0 / stride, points)
        # cosf / sinf of the device vs numpy may differ in the last bit of a sample position
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))
        gref = FO.feature_refine_backward(grad, boxes, 1.0 / stride, points)
        np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=0, atol=3e-5 * max(1.0, np.abs(gref).max()))


def test_feature_refine_module_shapes_and_grads(dev):
    from jdet_amd.ops.fr import FeatureRefineModule
    torch.manual_seed(0)
    m = FeatureRefineModule(16, [8, 16]).to(dev)
    rng = np.random.default_rng(0)
    feats = [torch.randn(2, 16, 16, 16, device=dev, requires_grad=True),
             torch.randn(2, 16, 8, 8, device=dev, requires_grad=True)]
    boxes = [[torch.from_numpy(_boxes(rng, 1, 16, 16, 8.0)[0].reshape(-1, 5)).to(dev),
              torch.from_numpy(_boxes(rng, 1, 8, 8, 16.0)[0].reshape(-1, 5)).to(dev)] for _ in range(2)]
    outs = m(feats, boxes)
    assert [tuple(o.shape) for o in outs] == [(2, 16, 16, 16), (2, 16, 8, 8)]
    sum(o.square().sum() for o in outs).backward()
    assert all(torch.isfinite(f.grad).all() and f.grad.abs().sum() > 0 for f in feats)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
