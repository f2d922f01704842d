"""OpenEarthMap readers (counterparts of dataset/oem.py of the reference; the fine-tune pair reader of dataset/oem_ft.py is
segland_amd/dataset/oem_ft.py).

The DataLoader workers only DECODE: a sample is the raw uint8 tile as rasterio returns it plus the random draws of the reference's
augmentation (made in the worker with the reference's generators, in the reference's order); `gpu_collate` / `TileAugmenter.prepare` then
crops, pads, flips, rotates, normalises and re-indexes the whole batch in one kernel launch on the GPU (csrc/augment.hip).

rasterio (GeoTIFF decoding) is NOT installed in the build image and is not stubbed: constructing a reader without it raises.  `--dataset
synthetic` is what the tests and benchmarks use; the kernel itself is pinned against the reference by golden G17 on synthetic tiles."""
import os
import os.path as osp
import random

import numpy as np
from torch.utils import data

from .augment import TileAugmenter, draw_train_params, remap_lut

BASE_CLASSES, NOVEL_CLASSES, NUM_CLASSES = set(range(1, 8)), set(range(8, 12)), 11          # oem.py:13,32,34
MEAN = STD = (0.5, 0.5, 0.5)                                                                   # oem.py:26-27


def _rasterio():
    try:
        import rasterio
        return rasterio
    except ImportError as e:
        raise RuntimeError('the OpenEarthMap readers need rasterio to decode GeoTIFF tiles (not installed in this environment); '
                           'use --dataset synthetic, or install rasterio') from e


def _read(root, sub, id_):
    return _rasterio().open(osp.join(root, sub, '%s.tif' % id_)).read()


class _Raw(data.Dataset):
    num_classes = NUM_CLASSES
    ignore_label = 255
    base_classes, novel_classes = BASE_CLASSES, NOVEL_CLASSES
    raw_tiles = True              # drivers: batches are lists of raw samples -> collate with `raw_collate`, prepare with `self.augmenter(device)`


class GFSSegTrain(_Raw):
    """dataset/oem.py:11-76.  __getitem__ -> (image uint8 [H,W,3], label uint8 [H,W], (h_off, w_off, flip, k), id)."""

    def __init__(self, root, list_path, fold, shot=1, mode='train', crop_size=(512, 512), ignore_label=255, base_size=(1024, 1024),
                 resize_label=False, filter=False, seed=123):
        _rasterio()
        assert mode in ('train', 'val_supp')
        self.root, self.crop_size, self.ignore_label, self.mode = root, tuple(crop_size), ignore_label, mode
        path = os.path.join(os.path.dirname(list_path), 'train.txt')
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.data_list = open(path).read().splitlines()

    def __len__(self):
        return len(self.novel_classes) if self.mode == 'val_supp' else len(self.data_list)        # oem.py:46-50

    def __getitem__(self, index):
        id_ = self.data_list[index]
        image = np.ascontiguousarray(np.rollaxis(_read(self.root, 'images', id_), 0, 3))
        label = np.ascontiguousarray(_read(self.root, 'labels', id_)[0])
        # mode 'val_supp': the same preparation with a centred crop (base_dataset.py:170-172); flip and rot90 are still drawn (oem.py:70-73)
        return image, label, draw_train_params(label, self.crop_size, self.ignore_label, self.mode), id_

    def augmenter(self, device):
        return TileAugmenter(self.crop_size, MEAN, STD, self.ignore_label, lut=None, device=device)


class GFSSegVal(_Raw):
    """dataset/oem.py:78-149 (resize_label=False path: full tiles, labels re-indexed to [bg | base 1..7 | novel 8..11])."""

    def __init__(self, root, list_path, fold, crop_size=(512, 512), ignore_label=255, base_size=(1024, 1024), resize_label=False,
                 use_novel=True, use_base=True):
        _rasterio()
        # resize_label=True (ft_pop.py:165): base_dataset.resize to base_size keeping the aspect ratio, then pad.  For tiles that already have
        # base_size (OpenEarthMap: 1024 x 1024 with --base-size 1024,1024) the scale factor is 1, cv2.resize returns the tile unchanged and nothing
        # is padded: that case is exact and supported.  Any other size needs cv2.resize's fixed-point INTER_LINEAR, which this build does not
        # restate (OpenCV is absent from the image, so it could not be pinned): __getitem__ raises for such a tile.
        self.resize_label, self.base_size = bool(resize_label), tuple(base_size)
        self.root, self.ignore_label, self.use_novel, self.use_base = root, ignore_label, use_novel, use_base
        self.ids = open(list_path).read().splitlines()

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, index):
        id_ = self.ids[index]
        image = np.ascontiguousarray(np.rollaxis(_read(self.root, 'images', id_), 0, 3))
        lp = osp.join(self.root, 'labels', '%s.tif' % id_)
        label = np.ascontiguousarray(_read(self.root, 'labels', id_)[0]) if os.path.exists(lp) else None
        if self.resize_label and label is not None and tuple(image.shape[:2]) != self.base_size:
            raise RuntimeError('GFSSegVal(resize_label=True): tile %s is %dx%d, base_size %dx%d -- only tiles that already have base_size are '
                               'supported (cv2.resize is not restated in this build)' % ((id_,) + tuple(image.shape[:2]) + self.base_size))
        return image, label, (0, 0, False, 0), id_

    def augmenter(self, device, size):
        return TileAugmenter(size, MEAN, STD, self.ignore_label, lut=remap_lut(self.base_classes, self.novel_classes, self.use_base, self.use_novel), device=device)


def raw_collate(batch):
    """Keep the variable-size raw tiles as lists (torch's default collate would try to stack them)."""
    return [b[:2] for b in batch], [b[2] for b in batch], [b[3] for b in batch]
