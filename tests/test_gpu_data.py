"""GPU: the device half of the input pipeline -- uint8 batches normalised on the device (csrc/image_prep.hip)."""
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _dataset(root, sizes, rng):
    os.makedirs(os.path.join(root, "images"))
    infos = []
    for i, (w, h) in enumerate(sizes):
        name = "P%04d.png" % i
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "images", name))
        infos.append(dict(filename=name, width=w, height=h, ann=dict(
            bboxes=np.array([[w / 2, h / 2, w / 4, h / 5, 0.3]], np.float32), labels=np.array([3], np.int64),
            bboxes_ignore=np.zeros((0, 5), np.float32), labels_ignore=np.zeros((0,), np.int64))))
    with open(os.path.join(root, "labels.pkl"), "wb") as f:
        pickle.dump(infos, f)


@pytest.mark.parametrize("to_bgr", [False, True])
def test_device_side_normalisation_equals_the_host_path(dev, tmp_path, to_bgr):
    """DeviceFeeder over Normalize(on_device=True) samples delivers bit-identical batches (same two float32 operations
    per element, zero padding to the batch canvas, channels-last) from a quarter of the host -> device bytes"""
    from jdet_amd.data import DeviceFeeder, DOTADataset
    rng = np.random.default_rng(8)
    root = str(tmp_path / "trainval")
    _dataset(root, [(96, 64), (64, 64), (80, 120), (120, 80)], rng)
    norm = dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=to_bgr)
    tfm = [dict(type="RotatedResize", min_size=64, max_size=128), dict(type="Pad", size_divisor=32)]
    host = DOTADataset(dataset_dir=root, transforms=tfm + [norm], batch_size=2)
    onde = DOTADataset(dataset_dir=root, transforms=tfm + [dict(norm, on_device=True)], batch_size=2)
    n = 0
    for (xh, th), (xd, td) in zip(DeviceFeeder(host.loader(), dev), DeviceFeeder(onde.loader(), dev)):
        assert xd.dtype == torch.float32 and xd.shape == xh.shape and xd.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(xd, xh)
        assert [t["filename"] for t in th] == [t["filename"] for t in td]
        n += 1
    assert n == 2
