"""An OpenEarthMap-shaped directory of synthetic TIFF tiles, read by the REAL readers of this package (dataset/oem.py, dataset/oem_ft.py): the
GeoTIFF decode, the list files and the class-list caches are exercised end to end (tests, tools/feed_rate.py, `--dataset synthetic_tiff`).

    <root>/images/<id>.tif   8-bit RGB        <root>/labels/<id>.tif   8-bit single band, OpenEarthMap numbering (0 unlabeled, 1..7 base, 8..11 novel)
    <root>/list/train.txt, val.txt, all_<shot>shot_seed<seed>.txt      (oem.py:38-42, oem_ft.py:57-69)

Tiles are written once per (root, parameters) by `make_dataset` (Pillow / libtiff; compression None, 'tiff_lzw', 'tiff_adobe_deflate' or 'packbits')."""
import os
import tempfile

import numpy as np
import torch

from . import oem, tiff

_MADE = {}


def tile_arrays(k, tile, seed, novel):
    """(image uint8 [H,W,3], label uint8 [H,W]) of synthetic tile k: smooth-ish images (compressible like aerial tiles are, unlike white noise), labels in
    32 x 32 blocks; `novel` tiles hold classes 8..11 too; every fourth tile starts with unlabeled rows."""
    g = torch.Generator().manual_seed(seed * 100003 + k)
    h, w = tile
    coarse_img = torch.randint(0, 256, ((h + 7) // 8, (w + 7) // 8, 3), generator=g, dtype=torch.uint8)
    img = coarse_img.repeat_interleave(8, 0).repeat_interleave(8, 1)[:h, :w]
    img = (img.to(torch.int16) + torch.randint(-6, 7, (h, w, 3), generator=g, dtype=torch.int16)).clamp(0, 255).to(torch.uint8).numpy()
    coarse = torch.randint(0, 12 if novel else 8, ((h + 31) // 32, (w + 31) // 32), generator=g)
    lab = coarse.repeat_interleave(32, 0).repeat_interleave(32, 1)[:h, :w].to(torch.uint8).numpy().copy()
    if k % 4 == 0:
        lab[: max(1, h // 10)] = 0
    return np.ascontiguousarray(img), lab


def make_dataset(root=None, n=24, tile=(608, 576), seed=123, shot=5, compression=None, n_val=8, repeat=1):
    """Writes the directory (once per parameter set in this process) and returns its root.  repeat: train.txt lists every tile that many times (a longer epoch over
    the same files: tools/feed_rate.py)."""
    key = (root, n, tuple(tile), seed, shot, compression, n_val, repeat)
    if key in _MADE and os.path.isdir(_MADE[key]):
        return _MADE[key]
    root = root or tempfile.mkdtemp(prefix='segland_synth_tiff_')
    for sub in ('images', 'labels', 'list'):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    ids = ['t%04d' % k for k in range(n)]
    val = ['v%04d' % k for k in range(n_val)]
    for j, id_ in enumerate(ids + val):
        img, lab = tile_arrays(j, tile, seed, novel=(j % 3 == 2))
        tiff.write_tiff(os.path.join(root, 'images', id_ + '.tif'), img, compression)
        tiff.write_tiff(os.path.join(root, 'labels', id_ + '.tif'), lab, compression)
    open(os.path.join(root, 'list', 'train.txt'), 'w').write(''.join(i + '\n' for i in ids * repeat))
    open(os.path.join(root, 'list', 'val.txt'), 'w').write(''.join(i + '\n' for i in val))
    novel = [i for k, i in enumerate(ids) if k % 3 == 2]
    open(os.path.join(root, 'list', 'all_%sshot_seed%s.txt' % (shot, seed)), 'w').write(''.join(i + '\n' for i in novel[:4 * shot]))
    _MADE[key] = root
    return root


def _root(root, crop_size, length, seed, shot, compression):
    if root and os.path.isdir(os.path.join(str(root), 'images')):
        return str(root)                                          # an existing directory of this layout (tools/feed_rate.py prepares one)
    return make_dataset(None, n=length, tile=(crop_size[0] + 96, crop_size[1] + 64), seed=seed, shot=shot, compression=compression)


class GFSSegTrain(oem.GFSSegTrain):
    """dataset/oem.py:11-76 of the reference on synthetic TIFF files (the driver's --data-dir / --train-list are ignored unless they point at such a directory)."""

    def __init__(self, root=None, list_path=None, fold=0, shot=5, mode='train', crop_size=(512, 512), ignore_label=255, base_size=(1024, 1024),
                 resize_label=False, filter=False, seed=123, length=24, compression=None, **kw):
        root = _root(root, crop_size, length, seed, shot, compression)
        super().__init__(root, os.path.join(root, 'list', 'train.txt'), fold, shot=shot, mode=mode, crop_size=crop_size, ignore_label=ignore_label,
                         base_size=base_size, resize_label=resize_label, filter=filter, seed=seed)


class GFSSegVal(oem.GFSSegVal):
    def __init__(self, root=None, list_path=None, fold=0, crop_size=(512, 512), ignore_label=255, base_size=(512, 512), resize_label=False,
                 use_novel=True, use_base=True, length=24, seed=123, shot=5, compression=None, **kw):
        root = root if (root and os.path.isdir(os.path.join(str(root), 'images'))) else \
            make_dataset(None, n=length, tile=tuple(base_size), seed=seed, shot=shot, compression=compression)
        super().__init__(root, os.path.join(root, 'list', 'val.txt'), fold, crop_size=crop_size, ignore_label=ignore_label, base_size=base_size,
                         resize_label=resize_label, use_novel=use_novel, use_base=use_base)
        self.tile = tuple(base_size)
