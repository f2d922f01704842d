"""Synthetic fine-tune pairs in the RAW sample format of dataset/oem_ft.py of this package: the list logic, the pair rule, `pair_collate`,
DataLoader workers and `PairAugmenter` (one GPU launch for the 2B tiles of a batch) run end to end without rasterio or list files."""
import os
import tempfile

import numpy as np
import torch

from .oem_ft import PairReader


class GFSSegTrain(PairReader):
    num_classes = 11

    def __init__(self, root=None, list_path=None, fold=0, shot=5, crop_size=(512, 512), base_size=(512, 512), mode='train', seed=123, filter=False,
                 length=24, tile=None, **kw):
        self.tile = tuple(tile) if tile else (crop_size[0] + 96, crop_size[1] + 64)
        self.seed = seed
        ids = ['s%03d' % k for k in range(length)]
        self._index = {i: k for k, i in enumerate(ids)}
        d = tempfile.mkdtemp(prefix='segland_synth_ft_')               # the reader caches its class lists next to the list file
        os.makedirs(os.path.join(d, 'list'))
        lst = os.path.join(d, 'list', 'train.txt')
        open(lst, 'w').write(''.join(i + '\n' for i in ids))
        novel = [i for i in ids if self._index[i] % 3 == 2]
        open(os.path.join(d, 'list', 'all_%sshot_seed%s.txt' % (shot, seed)), 'w').write(''.join(i + '\n' for i in novel[:4 * shot]))
        self._init_lists(lst, shot, mode, crop_size, 255, seed, False, True)

    def _coarse(self, id_):
        k = self._index[id_]
        g = torch.Generator().manual_seed(self.seed * 100003 + k)
        h, w = self.tile
        n_label = 12 if k % 3 == 2 else 8                               # every third tile holds novel classes
        return g, torch.randint(0, n_label, ((h + 31) // 32, (w + 31) // 32), generator=g)

    def read_label(self, id_):
        h, w = self.tile
        _, coarse = self._coarse(id_)
        return coarse.repeat_interleave(32, 0).repeat_interleave(32, 1)[:h, :w].to(torch.uint8).numpy().copy()

    def read_image(self, id_):
        h, w = self.tile
        g, _ = self._coarse(id_)
        return torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8).numpy()
