"""Synthetic fine-tune pairs with the interface of dataset/oem_ft.py:GFSSegTrain of the reference:
__getitem__ -> (img, mask, img_b, mask_b, cls): one novel-support tile (labels in {8..11, 255}: everything that is not
the novel class is ignored, oem_ft.py:197) and one base tile (labels 0..7)."""
import torch

from .synthetic import _Base


class GFSSegTrain(_Base):
    def __init__(self, root=None, list_path=None, fold=0, shot=5, crop_size=(512, 512), base_size=(512, 512), mode='train',
                 seed=123, filter=False, length=20, **kw):
        super().__init__(length, tuple(crop_size), 8, seed)
        self.novel_id_list = list(range(length))

    def update_base_list(self):
        self.seed += 1

    def __getitem__(self, i):
        img, mask = self._tile(i, lo=8, n_label=5)
        mask[mask == 12] = self.ignore_label
        img_b, mask_b = self._tile(i + 100000)
        mask_b[mask_b == self.ignore_label] = 0
        return img, mask, img_b, mask_b, 8 + i % 4
