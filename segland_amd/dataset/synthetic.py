"""Synthetic 8-channel-logit land-cover tiles with the interface of the reference's datasets
(dataset/oem.py:GFSSegTrain / GFSSegVal): __getitem__ -> (img [3,H,W] float32, mask [H,W] int64, id)."""
import torch
from torch.utils import data


class _Base(data.Dataset):
    num_classes = 11                       # OEM: 7 base + 4 novel (dataset/oem.py:13,32,34)
    base_classes = {1, 2, 3, 4, 5, 6, 7}
    novel_classes = {8, 9, 10, 11}
    ignore_label = 255

    def __init__(self, n, size, n_label, seed):
        self.n, self.size, self.n_label, self.seed = n, size, n_label, seed
        self.ids = list(range(n))
        self.data_list = self.ids

    def __len__(self):
        return self.n

    def _tile(self, i, lo=0, n_label=None):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        h, w = self.size
        img = torch.randn(3, h, w, generator=g).clamp_(-3, 3)
        coarse = torch.randint(lo, lo + (n_label or self.n_label), ((h + 31) // 32, (w + 31) // 32), generator=g)
        mask = coarse.repeat_interleave(32, 0).repeat_interleave(32, 1)[:h, :w].contiguous()
        mask[: max(1, h // 10)] = self.ignore_label if i % 4 == 0 else mask[: max(1, h // 10)]
        return img, mask


class GFSSegTrain(_Base):
    def __init__(self, root=None, list_path=None, fold=0, shot=1, crop_size=(512, 512), base_size=(512, 512), mode='train',
                 filter=False, length=64, seed=0, **kw):
        super().__init__(length, tuple(crop_size), 8, seed)

    def __getitem__(self, i):
        img, mask = self._tile(i)
        return img, mask, i


class GFSSegVal(_Base):
    def __init__(self, root=None, list_path=None, fold=0, base_size=(512, 512), resize_label=False, use_novel=False, use_base=True,
                 length=8, seed=1, **kw):
        super().__init__(length, tuple(base_size), 12 if use_novel else 8, seed)

    def __getitem__(self, i):
        img, mask = self._tile(i)
        return img, mask, i
