"""Fine-tune pairs (dataset/oem_ft.py of the reference) read by the real pair reader of this package from synthetic TIFF files: see dataset/synthetic_tiff.py."""
import os

from . import oem_ft
from .synthetic_tiff import _root


class GFSSegTrain(oem_ft.GFSSegTrain):
    def __init__(self, root=None, list_path=None, fold=0, shot=5, mode='train', crop_size=(512, 512), ignore_label=255, base_size=(1024, 1024),
                 resize_label=False, seed=123, filter=False, use_base=True, length=24, compression=None, **kw):
        root = _root(root, crop_size, length, seed, shot, compression)
        super().__init__(root, os.path.join(root, 'list', 'train.txt'), fold, shot=shot, mode=mode, crop_size=crop_size, ignore_label=ignore_label,
                         base_size=base_size, resize_label=resize_label, seed=seed, filter=filter, use_base=use_base)
