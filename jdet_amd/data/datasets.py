"""Datasets of the named configs and the loader that feeds the GPU.

Contract of python/jdet/data/custom.py:L14-119 (`CustomDataset`: `dataset_dir/{images, labels.pkl}` or explicit
`images_dir` + `annotations_file`; annotation records {'filename','width','height','ann':{'bboxes' (n,5),'labels',
'bboxes_ignore', ...}}; target dict keys L75-88; batches zero-padded to the largest image, L90-106),
data/dota.py:L22-87 (`DOTADataset`: class names by version, category balancing, `parse_result`), data/image.py:L13-119
(`ImageDataset`: images without ground truth).  `labels.pkl` is a pickle of plain python / numpy (what `jt.save`
writes), read with `pickle` here.

The reference's datasets are Jittor `Dataset`s that batch and prefetch themselves.  Here they are
`torch.utils.data.Dataset`s; `loader()` wraps them in a DataLoader (worker processes, the reference's collate rule,
pinned host memory) and `DeviceFeeder` overlaps the host-to-device copy of batch k+1 with the step on batch k on its
own HIP stream, delivering channels-last images and device-resident targets.  mAP evaluation (`voc_eval_dota` with
polygon IoU) lives in voc_eval.py: the IoU matrices come from the device kernel, the AP bookkeeping is numpy.
"""
import os
import pickle

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from jdet_amd.utils.registry import DATASETS

from .np_boxes import rotated_box_to_bbox_np, rotated_box_to_poly_np
from .transforms import Compose

DOTA1_CLASSES = ["plane", "baseball-diamond", "bridge", "ground-track-field", "small-vehicle", "large-vehicle", "ship",
                 "tennis-court", "basketball-court", "storage-tank", "soccer-ball-field", "roundabout", "harbor",
                 "swimming-pool", "helicopter"]
_DOTA_CLASSES = {"1": DOTA1_CLASSES, "1_5": DOTA1_CLASSES + ["container-crane"],
                 "2": DOTA1_CLASSES + ["container-crane", "airport", "helipad"]}
_IMG_EXT = (".jpg", ".bmp", ".jpeg", ".png", "tiff")


def _load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def collate_batch(batch):
    """[(image (3,h,w) float32, target)] -> ((N,3,Hmax,Wmax) float32 zero-padded at right / bottom, [targets]).
    Samples left un-normalised for the device (`Normalize(on_device=True)`: uint8 (h,w,3)) are collated as
    (N,Hmax,Wmax,3) uint8; each target then carries its image's extent in the canvas (`canvas_hw`)."""
    images, targets = zip(*batch)
    u8 = [im.dtype == np.uint8 and im.ndim == 3 and im.shape[-1] == 3 for im in images]
    if any(u8) and not all(u8):
        # a batch that mixes device-normalised (uint8 HWC) and host-normalised (float CHW) samples -- `Normalize(
        # on_device=True)` falls back per sample when an earlier transform left floats: finish the uint8 ones on the
        # host, so that the batch has ONE layout (deciding it from images[0] would collate the others wrongly)
        images, targets = list(images), [dict(t) for t in targets]
        for i, im in enumerate(images):
            if u8[i]:
                t = targets[i]
                arr = im.transpose((2, 0, 1)).astype(np.float32)
                if t.get("to_bgr"):
                    arr = arr[::-1]
                images[i] = ((arr - np.asarray(t["mean"], np.float32).reshape(-1, 1, 1)) /
                             np.asarray(t["std"], np.float32).reshape(-1, 1, 1)).astype(np.float32)
                t.pop("normalize_on_device", None)
        u8 = [False] * len(images)
    if all(u8):
        hmax = max(im.shape[0] for im in images)
        wmax = max(im.shape[1] for im in images)
        out = np.zeros((len(images), hmax, wmax, 3), dtype=np.uint8)
        targets = [dict(t) for t in targets]
        for i, im in enumerate(images):
            out[i, :im.shape[0], :im.shape[1]] = im
            targets[i]["canvas_hw"] = (int(im.shape[0]), int(im.shape[1]))
        return out, targets
    hmax = max(im.shape[-2] for im in images)
    wmax = max(im.shape[-1] for im in images)
    out = np.zeros((len(images), 3, hmax, wmax), dtype=np.float32)
    for i, im in enumerate(images):
        out[i, :, :im.shape[-2], :im.shape[-1]] = im
    return out, list(targets)


class _Base(Dataset):
    CLASSES = None

    def __init__(self, transforms, batch_size, num_workers, shuffle, drop_last=False):
        self.transforms = Compose(transforms) if isinstance(transforms, (list, tuple)) or transforms is None \
            else transforms
        self.batch_size, self.num_workers, self.shuffle, self.drop_last = batch_size, num_workers, shuffle, drop_last

    collate_batch = staticmethod(collate_batch)

    def loader(self, sampler=None, pin_memory=True):
        """DataLoader with the dataset's own batch size / workers / shuffle; pass a DistributedSampler for one
        process per GPU"""
        return DataLoader(self, batch_size=self.batch_size, shuffle=self.shuffle and sampler is None,
                          sampler=sampler, num_workers=self.num_workers, collate_fn=_collate_to_tensors,
                          pin_memory=pin_memory and torch.cuda.is_available(), drop_last=self.drop_last,
                          persistent_workers=self.num_workers > 0)


def _collate_to_tensors(batch):
    images, targets = collate_batch(batch)
    return torch.from_numpy(images), targets


@DATASETS.register_module()
class CustomDataset(_Base):
    def __init__(self, images_dir=None, annotations_file=None, dataset_dir=None, transforms=None, batch_size=1,
                 num_workers=0, shuffle=False, drop_last=False, filter_empty_gt=True, filter_min_size=-1,
                 buffer_size=512 * 1024 * 1024):
        super().__init__(transforms, batch_size, num_workers, shuffle, drop_last)
        if dataset_dir is not None:
            assert images_dir is None and annotations_file is None
            images_dir, annotations_file = os.path.join(dataset_dir, "images"), os.path.join(dataset_dir, "labels.pkl")
        else:
            assert images_dir is not None and annotations_file is not None
        self.images_dir = os.path.abspath(images_dir)
        self.annotations_file = os.path.abspath(annotations_file)
        self.img_infos = _load_pickle(self.annotations_file)
        if filter_empty_gt:
            self.img_infos = self._filter_imgs(filter_min_size)
        self.total_len = len(self.img_infos)

    def __len__(self):
        return self.total_len

    def _filter_imgs(self, min_size):
        return [info for info in self.img_infos
                if len(info["ann"]["bboxes"]) > 0 and min(info["width"], info["height"]) >= min_size]

    def _read_ann_info(self, idx):
        while len(self.img_infos[idx]["ann"]["bboxes"]) == 0:     # empty record: draw another one (L55-60)
            idx = int(np.random.choice(np.arange(self.total_len)))
        info = self.img_infos[idx]
        anno = info["ann"]
        img_path = os.path.join(self.images_dir, info["filename"])
        image = Image.open(img_path).convert("RGB")
        width, height = image.size
        assert width == info["width"] and height == info["height"], "image size is different from annotations"
        ignore = anno.get("bboxes_ignore", np.zeros((0, 5), np.float32))
        hboxes, polys = rotated_box_to_bbox_np(anno["bboxes"])
        hboxes_ignore, polys_ignore = rotated_box_to_bbox_np(ignore)
        return image, dict(
            rboxes=anno["bboxes"].astype(np.float32), hboxes=hboxes.astype(np.float32), polys=polys.astype(np.float32),
            labels=anno["labels"].astype(np.int32), rboxes_ignore=np.asarray(ignore, np.float32),
            hboxes_ignore=hboxes_ignore, polys_ignore=polys_ignore, classes=self.CLASSES,
            ori_img_size=(width, height), img_size=(width, height), scale_factor=1.0, filename=info["filename"],
            img_file=img_path)

    def __getitem__(self, idx):
        if "BATCH_IDX" in os.environ:
            idx = int(os.environ["BATCH_IDX"])
        image, anno = self._read_ann_info(idx)
        if self.transforms is not None:
            image, anno = self.transforms(image, anno)
        return image, anno

    def evaluate(self, results, work_dir, epoch, logger=None):
        raise NotImplementedError


@DATASETS.register_module()
class DOTADataset(CustomDataset):
    _BALANCE = {"storage-tank": (1, 526), "baseball-diamond": (2, 202), "ground-track-field": (1, 575),
                "swimming-pool": (2, 104), "soccer-ball-field": (1, 962), "roundabout": (1, 711),
                "tennis-court": (1, 655), "basketball-court": (4, 0), "helicopter": (8, 0), "container-crane": (50, 0)}

    def __init__(self, *args, balance_category=False, version="1", **kwargs):
        assert version in ["1", "1_5", "2"]
        self.CLASSES = _DOTA_CLASSES[version]
        super().__init__(*args, **kwargs)
        if balance_category:
            self.img_infos = self._balance_categories()
            self.total_len = len(self.img_infos)

    def _balance_categories(self):
        """rare classes repeated: every image holding class k appears l1 times, its first l2 images once more"""
        per_class = {}
        for idx, info in enumerate(self.img_infos):
            for label in np.unique(info["ann"]["labels"]):
                per_class.setdefault(int(label), []).append(idx)
        order = []
        for label, idxs in per_class.items():
            l1, l2 = self._BALANCE.get(self.CLASSES[label - 1], (1, 0))
            order.extend(idxs * l1 + idxs[:l2])
        return [self.img_infos[i] for i in order]

    def parse_result(self, results, save_path):
        """[((dets (k,6) [box, score], labels (k,)), image name)] -> one `<class>.txt` per class in DOTA's
        `name score x0 y0 ... x3 y3` submission format"""
        os.makedirs(save_path, exist_ok=True)
        lines = {}
        for (dets, labels), img_name in results:
            stem = os.path.splitext(img_name)[0]
            dets, labels = np.asarray(dets), np.asarray(labels)
            if dets.shape[0] == 0:
                continue
            polys = rotated_box_to_poly_np(dets[:, :5])
            for poly, score, label in zip(polys, dets[:, 5], labels):
                lines.setdefault(self.CLASSES[int(label)], []).append(
                    "{} {:.4f} ".format(stem, score) + " ".join("{:.4f}".format(v) for v in poly) + "\n")
        for classname, rows in lines.items():
            with open(os.path.join(save_path, classname + ".txt"), "w") as f:
                f.writelines(rows)

    def evaluate(self, results, work_dir, epoch, logger=None, save=True, iou_matrix_fn=None):
        """[((det_polys, det_scores, det_labels), target)] -> {"eval/<i>_<class>_AP", "eval/0_meanAP"}
        (data/dota.py:L89-146); IoU matrices on the HIP device unless `iou_matrix_fn` is given"""
        from .voc_eval import device_iou_matrix, evaluate_dota
        if save and work_dir is not None:
            path = os.path.join(work_dir, "detections/val_%s" % epoch)
            os.makedirs(path, exist_ok=True)
            with open(os.path.join(path, "val.pkl"), "wb") as f:
                pickle.dump(results, f)
        return evaluate_dota(results, self.CLASSES, iou_matrix_fn or device_iou_matrix)


@DATASETS.register_module()
class ImageDataset(_Base):
    _DEFAULT = [dict(type="Resize", min_size=[800], max_size=1333), dict(type="Pad", size_divisor=32),
                dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])]

    def __init__(self, images_file=None, images_dir="", dataset_type="DOTA", transforms=None, batch_size=1,
                 num_workers=0, shuffle=False):
        super().__init__(self._DEFAULT if transforms is None else transforms, batch_size, num_workers, shuffle)
        self.images_file = self._load_images(images_file, images_dir)
        self.total_len = len(self.images_file)
        self.dataset_type = dataset_type

    def __len__(self):
        return self.total_len

    @staticmethod
    def _load_images(images_file, images_dir):
        if not images_file:
            names = sorted(n for n in os.listdir(images_dir) if os.path.splitext(n)[1].lower() in _IMG_EXT)
        elif isinstance(images_file, list):
            names = list(images_file)
        elif isinstance(images_file, str):
            assert os.path.exists(images_file), f"{images_file} must be a file or list"
            names = []
            for rec in _load_pickle(images_file):
                if isinstance(rec, dict):
                    names.append(rec["filename"])
                elif isinstance(rec, str):
                    names.append(rec)
                else:
                    raise NotImplementedError
        else:
            raise NotImplementedError
        return [os.path.join(images_dir, n) for n in names]

    def __getitem__(self, index):
        if "BATCH_IDX" in os.environ:
            index = int(os.environ["BATCH_IDX"])
        img = Image.open(self.images_file[index]).convert("RGB")
        targets = dict(ori_img_size=img.size, img_size=img.size, scale_factor=1., img_file=self.images_file[index])
        if self.transforms:
            img, targets = self.transforms(img, targets)
        return img, targets


_TENSOR_KEYS = ("rboxes", "hboxes", "polys", "labels", "rboxes_ignore", "hboxes_ignore", "polys_ignore")


def targets_to_device(targets, device, non_blocking=True):
    """numpy target dicts -> the same dicts with the box / label arrays as device tensors (fp32 / int32)"""
    out = []
    for t in targets:
        d = dict(t)
        for k in _TENSOR_KEYS:
            if k in d and d[k] is not None:
                dt = torch.int32 if k == "labels" else torch.float32
                d[k] = torch.as_tensor(np.ascontiguousarray(d[k])).to(device=device, dtype=dt,
                                                                      non_blocking=non_blocking)
        out.append(d)
    return out


class DeviceFeeder:
    """Iterates a loader one batch ahead: the host-to-device copy of the next batch (pinned memory, its own HIP
    stream) runs while the caller computes on the current one; images arrive channels-last."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def _normalize(self, images_u8, targets):
        """uint8 (N,H,W,3) batch on the device -> normalised (N,3,H,W) float32, channels-last (one launch)"""
        from jdet_amd import _lib as L
        n, h, w, _ = images_u8.shape
        t0 = targets[0]
        for t in targets[1:]:   # one launch, one (mean, std, channel order): the batch must agree on them
            assert (np.array_equal(np.asarray(t["mean"]), np.asarray(t0["mean"]))
                    and np.array_equal(np.asarray(t["std"]), np.asarray(t0["std"]))
                    and bool(t["to_bgr"]) == bool(t0["to_bgr"])), "samples of one batch disagree on mean / std / to_bgr"
        valid = torch.tensor([t["canvas_hw"] for t in targets], dtype=torch.int32).to(images_u8.device,
                                                                                     non_blocking=True)
        out = torch.empty((n, h, w, 3), dtype=torch.float32, device=images_u8.device)
        L.check(L.lib().jdet_normalize_u8_nhwc(L.ptr(images_u8), L.ptr(valid), n, h, w,
                                               L.vecn(np.asarray(t0["mean"]).reshape(-1), 3),
                                               L.vecn(np.asarray(t0["std"]).reshape(-1), 3), int(bool(t0["to_bgr"])),
                                               L.ptr(out), L.stream_ptr(out)), "jdet_normalize_u8_nhwc")
        return out.permute(0, 3, 1, 2)

    def _stage(self, batch):
        images, targets = batch
        if images.dtype == torch.uint8:            # Normalize(on_device=True): a quarter of the bytes cross PCIe
            assert self.device.type == "cuda", "device-side normalisation needs a HIP device"
            with torch.cuda.stream(self.stream):
                images = self._normalize(images.to(self.device, non_blocking=True), targets)
                targets = targets_to_device(targets, self.device)
            return images, targets
        if self.stream is None:
            return images.to(self.device).contiguous(memory_format=torch.channels_last), \
                targets_to_device(targets, self.device, False)
        with torch.cuda.stream(self.stream):
            images = images.to(self.device, non_blocking=True).contiguous(memory_format=torch.channels_last)
            targets = targets_to_device(targets, self.device)
        return images, targets

    def __iter__(self):
        it = iter(self.loader)
        nxt = None
        for batch in it:
            staged = self._stage(batch)
            if nxt is not None:
                yield self._ready(nxt)
            nxt = staged
        if nxt is not None:
            yield self._ready(nxt)

    def _ready(self, staged):
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            for t in [staged[0]] + [v for d in staged[1] for v in d.values() if torch.is_tensor(v)]:
                t.record_stream(torch.cuda.current_stream(self.device))
        return staged

    def __len__(self):
        return len(self.loader)
