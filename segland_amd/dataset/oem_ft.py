"""OpenEarthMap fine-tune PAIR reader (counterpart of dataset/oem_ft.py of the reference).

`GFSSegTrain` keeps the reference's list logic -- class -> tile-id lists (`_filter_and_map_ids`, oem_ft.py:262-299, cached in
`train_base_class<c>.txt` like the reference does), the per-class `shot` base tiles (`_get_supp_list` :126-181 == `update_base_list`
:72-124, same draws from `random` in the same order) and the pair rule of `_get_train_sample` (:189-220): a RANDOM novel support tile whose
unlabeled pixels become ignore BEFORE the crop is drawn (:197), the index-th base tile, and two independent crop / flip / rot90 draw sets,
novel tile first.  What differs is where the pixels are prepared: the DataLoader worker only decodes and draws; crop, pad, flip, rot90,
normalisation and the int64 labels of BOTH tiles of every pair of a batch are ONE launch of csrc/augment.hip (`PairAugmenter`).

Two reference behaviours kept on purpose: the reader never overrides BaseDataset's ImageNet mean / std (base_dataset.py:9; oem.py:26-27
sets 0.5 / 0.5 for base training only), and mode='train' with use_base=False has no base list (the reference raises AttributeError in
__len__; here the constructor says so).  mode='val_supp' (`_get_val_support`, cv2.warpAffine rotations) is not on the ft_pop path and raises.
rasterio is needed to decode GeoTIFF and is not installed in the build image: `--dataset synthetic_raw` drives the same sample format."""
import os
import os.path as osp
import random
from collections import defaultdict

import numpy as np

from .augment import TileAugmenter, draw_train_params
from .oem import BASE_CLASSES, NOVEL_CLASSES, NUM_CLASSES, PackedTiles, _Raw, _decoder, _read, crop_rows

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)               # base_dataset.py:9 (not overridden by oem_ft.py)


class PairAugmenter:
    """prepare(pairs, params) -> (img, mask, img_b, mask_b) on the GPU: the 2B tiles of a batch of pairs go through one sl_augment_batch launch."""

    def __init__(self, crop_size, mean=MEAN, std=STD, ignore_label=255, device='cuda'):
        self.aug = TileAugmenter(crop_size, mean, std, ignore_label, lut=None, device=device)

    def prepare(self, pairs, params):
        """pairs: PackedTiles from PairCollate (2B tiles, novel / base alternating) or a list of ((novel image, label), (base image, label))."""
        B = len(params)
        prm = [q[0] for q in params] + [q[1] for q in params]
        if isinstance(pairs, PackedTiles):
            img, lab = self.aug.prepare(pairs, prm, order=list(range(0, 2 * B, 2)) + list(range(1, 2 * B, 2)))
        else:
            img, lab = self.aug.prepare([p[0] for p in pairs] + [p[1] for p in pairs], prm)
        return img[:B], lab[:B], img[B:], lab[B:]


class PairCollate:
    """(PackedTiles of the 2B tiles in the order novel_0, base_0, novel_1, base_1, ..., [(novel draws, base draws)], [ids]): like oem.RawCollate the tiles are cut
    down to the rows their crops read and leave the worker as one shared-memory buffer.  `pairs[2 * i]` / `pairs[2 * i + 1]` are pair i's novel / base tile."""

    def __init__(self, crop_h=None):
        self.crop_h = crop_h

    def __call__(self, batch):
        tiles, params = [], []
        for (nov, base), (pn, pb), _ in batch:
            ni, nl, pn = crop_rows(nov[0], nov[1], pn, self.crop_h)
            bi, bl, pb = crop_rows(base[0], base[1], pb, self.crop_h)
            tiles += [(ni, nl), (bi, bl)]
            params.append((pn, pb))
        return PackedTiles(tiles), params, [b[2] for b in batch]


pair_collate = PairCollate()


class PairReader(_Raw):
    """The list logic and the pair rule, independent of how a tile is decoded (`read_image(id) -> uint8 [H,W,3]`, `read_label(id) -> uint8 [H,W]`)."""
    pair_tiles = True
    collate_fn = staticmethod(pair_collate)

    def _init_lists(self, list_path, shot, mode, crop_size, ignore_label, seed, filter, use_base):
        if mode == 'val_supp':
            raise RuntimeError("oem_ft.GFSSegTrain(mode='val_supp') rotates support tiles with cv2.warpAffine (oem_ft.py:247, base_dataset.py:117-132); "
                               'it is not on the ft_pop path and is not built')
        assert mode == 'train'
        self.list_path, self.shot, self.mode, self.crop_size, self.ignore_label, self.use_base = list_path, shot, mode, tuple(crop_size), ignore_label, use_base
        self.collate_fn = PairCollate(self.crop_size[0])
        self.base_classes, self.novel_classes = set(BASE_CLASSES), set(NOVEL_CLASSES)
        filter_flag = bool(filter)
        list_dir = os.path.dirname(list_path) + ('_filter' if filter_flag else '')
        first = list(self.base_classes)[0]
        if os.path.exists(os.path.join(list_dir, 'train_base_class%s.txt' % first)):
            self.base_cls_to_ids = defaultdict(list)
            for cls in self.base_classes:
                self.base_cls_to_ids[cls] = open(os.path.join(list_dir, 'train_base_class%s.txt' % cls)).read().splitlines()
        else:
            self.ids = open(list_path).read().splitlines()
            self.base_cls_to_ids, self.novel_cls_to_ids = self._filter_and_map_ids(filter_flag)
            for cls in self.base_classes:
                with open(os.path.join(list_dir, 'train_base_class%s.txt' % cls), 'w') as f:
                    f.writelines(i + '\n' for i in self.base_cls_to_ids[cls])
        self.novel_id_list = open(os.path.join(list_dir, 'all_%sshot_seed%s.txt' % (shot, seed))).read().splitlines()
        if not use_base:
            raise RuntimeError("oem_ft.GFSSegTrain(mode='train', use_base=False) has no base tile list (the reference's __len__ raises AttributeError)")
        self.update_base_list()

    def _filter_and_map_ids(self, filter_intersection=False):
        base_cls_to_ids, novel_cls_to_ids = defaultdict(list), defaultdict(list)
        for id_ in self.ids:
            present = set(np.unique(self.read_label(id_)).tolist())
            labelled = present - {0}
            if not filter_intersection or labelled <= self.base_classes:      # filtered lists: tiles without any novel class only
                for cls in present & self.base_classes:
                    base_cls_to_ids[cls].append(id_)
            for cls in present & self.novel_classes:
                novel_cls_to_ids[cls].append(id_)
        return base_cls_to_ids, novel_cls_to_ids

    def update_base_list(self):
        """A fresh draw of `shot` tiles per base class (also what the constructor's _get_supp_list does).  The reference opens every drawn label
        again only to PRINT how many base tiles hold novel classes; no draw depends on it, so it is not read here."""
        base_id_list = []
        for cls in list(self.base_classes):
            files = self.base_cls_to_ids[cls]
            n = len(files)
            if n == 0 and self.shot > 0:
                raise RuntimeError('no training tile holds base class %s' % cls)
            if n < self.shot:
                base_id_list += list(files)
                base_id_list += [files[random.randint(1, n) - 1] for _ in range(self.shot - n)]
            else:
                base_id_list += [files[j] for j in random.choices(list(range(n)), k=self.shot)]
        self.base_id_list = base_id_list
        self.supp_cls_id_list = self.novel_id_list + base_id_list

    def __len__(self):
        return len(self.base_id_list)

    def __getitem__(self, index):
        """-> ((novel tile, base tile), (novel draws, base draws), novel id); a tile = (image uint8 [H,W,3], label uint8 [H,W])."""
        id_b = self.base_id_list[index]
        id_ = random.choice(self.novel_id_list)
        image, label = self.read_image(id_), self.read_label(id_)
        label = np.where(label == 0, np.uint8(self.ignore_label), label).astype(np.uint8)       # oem_ft.py:197, before the crop draw looks at the labels
        image_b, label_b = self.read_image(id_b), self.read_label(id_b)
        prm = draw_train_params(label, self.crop_size, self.ignore_label)
        prm_b = draw_train_params(label_b, self.crop_size, self.ignore_label)
        return ((image, label), (image_b, label_b)), (prm, prm_b), id_

    def augmenter(self, device):
        return PairAugmenter(self.crop_size, MEAN, STD, self.ignore_label, device=device)


class GFSSegTrain(PairReader):
    num_classes = NUM_CLASSES

    def __init__(self, root, list_path, fold, shot=1, mode='train', crop_size=(512, 512), ignore_label=255, base_size=(1024, 1024),
                 resize_label=False, seed=123, filter=False, use_base=True):
        _decoder()
        self.root = root
        self._init_lists(list_path, shot, mode, crop_size, ignore_label, seed, filter, use_base)

    def read_image(self, id_):
        return np.ascontiguousarray(np.rollaxis(_read(self.root, 'images', id_), 0, 3))

    def read_label(self, id_):
        return np.ascontiguousarray(_read(self.root, 'labels', id_)[0])
