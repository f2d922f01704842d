"""OpenEarthMap readers (counterparts of dataset/oem.py of the reference; the fine-tune pair reader of dataset/oem_ft.py is
segland_amd/dataset/oem_ft.py).

The DataLoader workers only DECODE: a sample is the raw uint8 tile as rasterio returns it plus the random draws of the reference's
augmentation (made in the worker with the reference's generators, in the reference's order); `raw_collate` / `TileAugmenter.prepare` then
crops, pads, flips, rotates, normalises and re-indexes the whole batch in one kernel launch on the GPU (csrc/augment.hip).

Decode: dataset/tiff.py -- rasterio when installed, else Pillow (8-bit RGB / single-band TIFF, uncompressed or LZW / Deflate / PackBits).  `--dataset
synthetic_tiff` writes an OpenEarthMap-shaped directory of synthetic TIFF tiles and runs THESE readers on it (tests, tools/feed_rate.py); `--dataset
synthetic_raw` skips the decode.  The preparation kernel is pinned against the reference by golden G17."""
import os
import os.path as osp
import random

import numpy as np
from torch.utils import data

from . import tiff
from .augment import TileAugmenter, draw_train_params, remap_lut

BASE_CLASSES, NOVEL_CLASSES, NUM_CLASSES = set(range(1, 8)), set(range(8, 12)), 11          # oem.py:13,32,34
MEAN = STD = (0.5, 0.5, 0.5)                                                                   # oem.py:26-27


def _decoder():
    """Raises when no TIFF decoder is installed (a reader is useless then); returns its name."""
    return tiff.backend()


def _read(root, sub, id_):
    """uint8 [bands, H, W] like rasterio.open(...).read() (dataset/oem.py:57-58 of the reference)."""
    return tiff.read_tiff(osp.join(root, sub, '%s.tif' % id_))


class _Raw(data.Dataset):
    num_classes = NUM_CLASSES
    ignore_label = 255
    base_classes, novel_classes = BASE_CLASSES, NOVEL_CLASSES
    raw_tiles = True              # drivers: batches are lists of raw samples -> collate with `raw_collate`, prepare with `self.augmenter(device)`


class GFSSegTrain(_Raw):
    """dataset/oem.py:11-76.  __getitem__ -> (image uint8 [H,W,3], label uint8 [H,W], (h_off, w_off, flip, k), id)."""

    def __init__(self, root, list_path, fold, shot=1, mode='train', crop_size=(512, 512), ignore_label=255, base_size=(1024, 1024),
                 resize_label=False, filter=False, seed=123):
        _decoder()
        assert mode in ('train', 'val_supp')
        self.root, self.crop_size, self.ignore_label, self.mode = root, tuple(crop_size), ignore_label, mode
        path = os.path.join(os.path.dirname(list_path), 'train.txt')
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.data_list = open(path).read().splitlines()
        self.collate_fn = RawCollate(self.crop_size[0])

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
        _decoder()
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


def crop_rows(img, lab, prm, crop_h):
    """Only the rows the crop will read: a tile at least as tall as the crop is never padded vertically (base_dataset.py:88-104 pads tiles SMALLER than the crop),
    so rows [h_off, h_off + crop_h) -- a contiguous slice of the row-major tile -- are all the preparation kernel looks at; the draw's h_off becomes 0."""
    h_off, w_off, flip, k = prm
    if crop_h and img.shape[0] > crop_h and 0 <= h_off <= img.shape[0] - crop_h:
        img = img[h_off:h_off + crop_h]
        lab = None if lab is None else lab[h_off:h_off + crop_h]
        prm = (0, w_off, flip, k)
    return img, lab, prm


class PackedTiles:
    """The raw tiles of a batch in ONE uint8 tensor: it crosses the process boundary as a single shared-memory segment (one file descriptor, one mmap in the trainer
    instead of two per tile) and goes to the GPU as a single copy; `meta[i]` = (image byte offset, H, W, label byte offset or -1).  Behaves like the list of
    (image uint8 [H,W,3], label uint8 [H,W] | None) it replaces (len, indexing, iteration: views into the buffer)."""

    def __init__(self, tiles):
        import torch
        meta, off = [], 0
        for img, lab in tiles:
            H, W = int(img.shape[0]), int(img.shape[1])
            lo = -1
            io, off = off, off + (H * W * 3 + 15) // 16 * 16
            if lab is not None:
                lo, off = off, off + (H * W + 15) // 16 * 16
            meta.append((io, H, W, lo))
        self.meta = meta
        self.buf = torch.empty(max(off, 16), dtype=torch.uint8)
        flat = self.buf.numpy()
        for (img, lab), (io, H, W, lo) in zip(tiles, meta):
            flat[io:io + H * W * 3] = np.asarray(img).reshape(-1)
            if lo >= 0:
                flat[lo:lo + H * W] = np.asarray(lab).reshape(-1)

    def __len__(self):
        return len(self.meta)

    def __getitem__(self, i):
        io, H, W, lo = self.meta[i]
        return self.buf[io:io + H * W * 3].view(H, W, 3), (None if lo < 0 else self.buf[lo:lo + H * W].view(H, W))

    def __iter__(self):
        return (self[i] for i in range(len(self.meta)))


class RawCollate:
    """Keeps the variable-size raw tiles of a batch apart (torch's default collate would try to stack them): (PackedTiles, [draws], [ids]).  Runs in the worker:
    the tiles are cut down to the rows their crops read when `crop_h` is given (training readers) and leave the worker as one shared-memory buffer."""

    def __init__(self, crop_h=None):
        self.crop_h = crop_h

    def __call__(self, batch):
        tiles, params = [], []
        for img, lab, prm, _ in batch:
            img, lab, prm = crop_rows(img, lab, prm, self.crop_h)
            tiles.append((img, lab))
            params.append(prm)
        return PackedTiles(tiles), params, [b[3] for b in batch]


raw_collate = RawCollate()
