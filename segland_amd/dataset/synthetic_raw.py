"""Synthetic tiles in the RAW form of the OpenEarthMap readers (dataset/oem.py of this package): uint8 [H,W,3] images and uint8 labels in the
dataset's own class numbering, larger than the crop, with the reference's random draws -- the driver path with DataLoader workers +
`raw_collate` + `TileAugmenter` (one GPU launch per batch) runs end to end without rasterio."""
import numpy as np
import torch

from .augment import TileAugmenter, draw_train_params, remap_lut
from .oem import MEAN, STD, RawCollate, _Raw, raw_collate  # noqa: F401


class _Base(_Raw):
    def __init__(self, n, tile, seed):
        self.n, self.tile, self.seed = n, tile, seed
        self.ids = self.data_list = list(range(n))

    def __len__(self):
        return self.n

    def _tile(self, i, n_label):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        h, w = self.tile
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8).numpy()
        coarse = torch.randint(0, n_label, ((h + 31) // 32, (w + 31) // 32), generator=g)
        lab = coarse.repeat_interleave(32, 0).repeat_interleave(32, 1)[:h, :w].to(torch.uint8).numpy().copy()
        if i % 4 == 0:
            lab[: max(1, h // 10)] = 255
        return img, lab


class GFSSegTrain(_Base):
    def __init__(self, root=None, list_path=None, fold=0, shot=1, crop_size=(512, 512), base_size=(512, 512), mode='train', filter=False,
                 length=64, seed=0, **kw):
        super().__init__(length, (crop_size[0] + 96, crop_size[1] + 64), seed)
        self.crop_size = tuple(crop_size)
        self.collate_fn = RawCollate(self.crop_size[0])

    def __getitem__(self, i):
        img, lab = self._tile(i, 8)
        return img, lab, draw_train_params(lab, self.crop_size, self.ignore_label), i

    def augmenter(self, device):
        return TileAugmenter(self.crop_size, MEAN, STD, self.ignore_label, device=device)


class GFSSegVal(_Base):
    def __init__(self, root=None, list_path=None, fold=0, base_size=(512, 512), resize_label=False, use_novel=False, use_base=True, length=8,
                 seed=1, **kw):
        super().__init__(length, tuple(base_size), seed)
        self.use_novel, self.use_base = use_novel, use_base

    def __getitem__(self, i):
        img, lab = self._tile(i, 12 if self.use_novel else 8)
        return img, lab, (0, 0, False, 0), i

    def augmenter(self, device, size=None):
        return TileAugmenter(size or self.tile, MEAN, STD, self.ignore_label, lut=remap_lut(self.base_classes, self.novel_classes, self.use_base, self.use_novel), device=device)
