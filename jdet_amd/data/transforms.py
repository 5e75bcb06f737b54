"""Image / target transforms of the named configs.  Contract of python/jdet/data/transforms.py: `Compose` L12-29,
`Resize` L79-152, `RotatedResize` L315-342, `RandomFlip` L344-387, `RotatedRandomFlip` L389-441, `Pad` L443-465,
`Normalize` L467-487 -- PIL image in, (3,H,W) float32 array out of `Normalize`; targets are dicts of numpy arrays
with the keys of data/custom.py:L75-88; sizes are stored (width, height) as PIL reports them."""
import random

import numpy as np
from PIL import Image

from jdet_amd.utils.registry import TRANSFORMS, build_from_cfg

from .np_boxes import norm_angle, poly_to_rotated_box_np, rotated_box_to_poly_np

_BOX_KEYS = ("bboxes", "hboxes", "rboxes", "polys", "hboxes_ignore", "polys_ignore", "rboxes_ignore")


@TRANSFORMS.register_module()
class Compose:
    def __init__(self, transforms=None):
        self.transforms = []
        for t in transforms or []:
            if isinstance(t, dict):
                t = build_from_cfg(t, TRANSFORMS)
            elif not callable(t):
                raise TypeError("transform must be callable or a dict")
            self.transforms.append(t)

    def __call__(self, image, target=None):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


@TRANSFORMS.register_module()
class Resize:
    def __init__(self, min_size, max_size, keep_ratio=True):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size
        self.keep_ratio = keep_ratio

    def get_size(self, image_size):
        """-> ((out_h, out_w), scale).  Short side to a randomly chosen `min_size`, limited to [1/1.5, 1.5] x the
        original short side and so that the long side stays <= max_size (L88-126)."""
        w, h = image_size
        if not self.keep_ratio:
            oh, ow = self.min_size[0], self.max_size
            return (oh, ow), oh / h
        short, long_ = (w, h) if w <= h else (h, w)
        size = int(np.clip(random.choice(self.min_size), int(short / 1.5), int(short * 1.5)))
        if self.max_size is not None and float(long_) / float(short) * size > self.max_size:
            size = int(round(self.max_size * float(short) / float(long_)))
        if short == size:
            return (h, w), 1.
        if w < h:
            ow, oh = size, int(size * h / w)
        else:
            oh, ow = size, int(size * w / h)
        assert abs(oh / h - ow / w) < 1e-2
        return (oh, ow), oh / h

    _keys = ("bboxes", "polys")

    def _scale_clip(self, boxes, old_size, new_size):
        (w0, h0), (w1, h1) = old_size, new_size
        boxes = boxes.copy()
        boxes[:, 0::2] = np.clip(boxes[:, 0::2] * float(w1 / w0), 0, w1 - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2] * float(h1 / h0), 0, h1 - 1)
        return boxes

    def _resize_boxes(self, target, size):
        for key in self._keys:
            if key in target:
                target[key] = self._scale_clip(target[key], target["img_size"], size)

    def __call__(self, image, target=None):
        size, scale_factor = self.get_size(image.size)
        image = image.resize(size[::-1], Image.BILINEAR)
        if target is not None:
            self._resize_boxes(target, image.size)
            target["img_size"] = image.size
            target["scale_factor"] = scale_factor
            target["pad_shape"] = image.size
            target["keep_ratio"] = self.keep_ratio
        return image, target


@TRANSFORMS.register_module()
class RotatedResize(Resize):
    """rotated boxes go through their polygons: scale + clip the vertices, fit the box back (L317-342)"""
    _keys = _BOX_KEYS

    def _resize_boxes(self, target, size):
        for key in self._keys:
            boxes = target.get(key)
            if boxes is None or boxes.ndim != 2:
                continue
            if "rboxes" in key:
                target[key] = poly_to_rotated_box_np(self._scale_clip(rotated_box_to_poly_np(boxes),
                                                                      target["img_size"], size))
            else:
                target[key] = self._scale_clip(boxes, target["img_size"], size)


def _flip_hboxes(boxes, w, h, direction):
    out = boxes.copy()
    if direction in ("horizontal", "diagonal"):
        out[..., 0::4] = w - boxes[..., 2::4]
        out[..., 2::4] = w - boxes[..., 0::4]
    if direction in ("vertical", "diagonal"):
        out[..., 1::4] = h - boxes[..., 3::4]
        out[..., 3::4] = h - boxes[..., 1::4]
    return out


@TRANSFORMS.register_module()
class RandomFlip:
    _keys = ("bboxes", "polys")

    def __init__(self, prob=0.5, direction="horizontal"):
        assert direction in ["horizontal", "vertical", "diagonal"], f"{direction} not supported"
        self.direction = direction
        self.prob = prob

    def _flip_boxes(self, target, size):
        w, h = target["img_size"]
        for key in self._keys:
            if key in target:
                target[key] = _flip_hboxes(target[key], w, h, self.direction)

    def _flip_image(self, image):
        if self.direction in ("horizontal", "diagonal"):
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
        if self.direction in ("vertical", "diagonal"):
            image = image.transpose(Image.FLIP_TOP_BOTTOM)
        return image

    def __call__(self, image, target=None):
        if random.random() < self.prob:
            image = self._flip_image(image)
            if target is not None:
                self._flip_boxes(target, image.size)
            target["flip"] = self.direction
        return image, target


@TRANSFORMS.register_module()
class RotatedRandomFlip(RandomFlip):
    """centres and polygon vertices mirror as w - x - 1 (pixel centres), horizontal boxes as w - x (L391-441)"""
    _keys = _BOX_KEYS

    def _flip_rboxes(self, boxes, w, h):
        out = boxes.copy()
        if self.direction == "horizontal":
            out[..., 0::5] = w - out[..., 0::5] - 1
            out[..., 4::5] = norm_angle(np.pi - out[..., 4::5])
        elif self.direction == "vertical":
            out[..., 1::5] = h - out[..., 1::5] - 1
            out[..., 4::5] = norm_angle(-out[..., 4::5])
        else:
            assert False, "rotated boxes: horizontal or vertical flips only"
        return out

    def _flip_polys(self, polys, w, h):
        out = polys.copy()
        if self.direction in ("horizontal", "diagonal"):
            out[..., 0::2] = w - out[..., 0::2] - 1
        if self.direction in ("vertical", "diagonal"):
            out[..., 1::2] = h - out[..., 1::2] - 1
        return out

    def _flip_boxes(self, target, size):
        w, h = size
        for key in self._keys:
            if key not in target:
                continue
            if "rboxes" in key:
                target[key] = self._flip_rboxes(target[key], w, h)
            elif "polys" in key:
                target[key] = self._flip_polys(target[key], w, h)
            else:
                target[key] = _flip_hboxes(target[key], w, h, self.direction)


@TRANSFORMS.register_module()
class Pad:
    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert (size is None) != (size_divisor is None), "exactly one of size / size_divisor"
        self.size = size
        self.size_divisor = size_divisor
        self.pad_val = pad_val

    def __call__(self, image, target=None):
        if self.size is not None:
            pad_w, pad_h = self.size
        else:
            d = self.size_divisor
            pad_w, pad_h = int(np.ceil(image.size[0] / d)) * d, int(np.ceil(image.size[1] / d)) * d
        canvas = Image.new(image.mode, (pad_w, pad_h), (self.pad_val,) * len(image.split()))
        canvas.paste(image, (0, 0, image.size[0], image.size[1]))
        target["pad_shape"] = canvas.size
        return canvas, target


@TRANSFORMS.register_module()
class Normalize:
    """`on_device=True` (not in the reference): the arithmetic is left to the device -- the sample stays the uint8
    (H, W, 3) array the decoder produced, `collate_batch` pads uint8, and `DeviceFeeder` uploads a quarter of the
    bytes and runs jdet_normalize_u8_nhwc (the same two float32 operations per element: bit-identical images)."""

    def __init__(self, mean, std, to_bgr=True, on_device=False):
        self.mean = np.float32(mean).reshape(-1, 1, 1)
        self.std = np.float32(std).reshape(-1, 1, 1)
        self.to_bgr = to_bgr
        self.on_device = on_device

    def __call__(self, image, target=None):
        target["mean"] = self.mean
        target["std"] = self.std
        target["to_bgr"] = self.to_bgr
        if self.on_device:
            arr = np.array(image) if isinstance(image, Image.Image) else np.asarray(image).transpose((1, 2, 0))
            if arr.dtype == np.uint8:
                target["normalize_on_device"] = True
                return np.ascontiguousarray(arr), target
            # an earlier transform left floats (or anything that is not the decoder's uint8): casting would wrap or
            # truncate silently -- normalise this sample on the host instead
            image = arr.transpose((2, 0, 1))
        if isinstance(image, Image.Image):
            image = np.array(image).transpose((2, 0, 1))
        if self.to_bgr:
            image = image[::-1]
        image = (image - self.mean) / self.std
        return image, target
