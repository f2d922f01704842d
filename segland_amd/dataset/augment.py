"""GPU tile preparation for the OpenEarthMap readers (counterpart of dataset/base_dataset.py:29-175 of the reference).

The reference prepares every tile on DataLoader worker CPUs (numpy + OpenCV); at >= 500 tiles/s per GPU that path starves the training step.
Here the workers only DECODE (rasterio) and hand over the raw uint8 tile; crop / pad / flip / rot90 / channel reversal / normalisation /
label re-indexing of a whole batch is one kernel launch (csrc/augment.hip).  The random draws are made on the host with the reference's
generators and in the reference's order, so a seeded run picks the same crops."""
import ctypes as C
import random
import struct

import numpy as np
import torch

from .. import _lib
from ..ops import _p, _s


def draw_train_params(label, crop_size, ignore_label=255, mode='train'):
    """(h_off, w_off, flip, k): base_dataset.py:140-155 (np.random crop offsets, redrawn while the crop is all-ignore; outside train mode the
    crop is centred and nothing is drawn for it, :170-172), then :106-110 and :134-138 (one random.random() each for flip and rot90 count)."""
    H, W = label.shape
    ch, cw = crop_size
    mh, mw = max(H - ch, 0), max(W - cw, 0)
    if mode == 'train':
        while True:
            h_off, w_off = np.random.randint(0, mh + 1), np.random.randint(0, mw + 1)
            u = np.unique(label[h_off:h_off + ch, w_off:w_off + cw])
            if not (u.size == 1 and int(u[0]) == ignore_label):
                break
    else:
        h_off, w_off = int(round(mh / 2.)), int(round(mw / 2.))
    flip = random.random() < 0.5
    k = int(random.random() // 0.25)
    return h_off, w_off, flip, k


def remap_lut(base_classes, novel_classes, use_base=True, use_novel=True):
    """dataset/oem.py:113-133 as a lookup table (uint8[256])."""
    lut = np.arange(256, dtype=np.uint8)
    base, novel = list(base_classes), list(novel_classes)
    for c in base:
        lut[c] = base.index(c) + 1 if use_base else 0
    for c in novel:
        lut[c] = (novel.index(c) + (len(base) + 1 if use_base else 1)) if use_novel else 0
    return lut


class TileAugmenter:
    """prepare(tiles, params) -> (image [B,3,ch,cw] float32, label [B,ch,cw] int64) on the GPU.
    tiles: list of (image uint8 [H,W,3], label uint8 [H,W] or None) numpy arrays or CPU tensors (what the readers' collate functions hand over: shared-memory
    tensors, dataset/oem.py RawCollate); params: list of (h_off, w_off, flip, k)."""

    def __init__(self, crop_size, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), ignore_label=255, lut=None, device='cuda'):
        self.crop_size, self.ignore_label, self.device = tuple(crop_size), ignore_label, torch.device(device)
        self.mean, self.std = (C.c_double * 3)(*mean), (C.c_double * 3)(*std)      # float64 like the python lists numpy broadcasts in normalize()
        self.lut = None if lut is None else torch.as_tensor(np.asarray(lut, dtype=np.uint8)).to(self.device)

    def prepare(self, tiles, params, order=None):
        """order: tile indices in output order (PairAugmenter: all novel tiles, then all base tiles); default 0..B-1."""
        ch, cw = self.crop_size
        if any(k % 2 for _, _, _, k in params) and ch != cw:
            raise ValueError('rot90 by an odd count needs square crops')
        from .oem import PackedTiles
        B = len(params)
        order = list(range(B)) if order is None else list(order)
        keep, rec, flags = [], b'', []
        if isinstance(tiles, PackedTiles):
            # one host -> device copy for the whole batch (the collate in the worker packed it, already cut down to the crops' rows)
            dbuf = tiles.buf.to(self.device, non_blocking=True)
            base = dbuf.data_ptr()
            keep.append(dbuf)
            with_label = tiles.meta[order[0]][3] >= 0
            for i, (h_off, w_off, flip, k) in zip(order, params):
                io, H, W, lo = tiles.meta[i]
                rec += struct.pack('<QQiiii', base + io, 0 if lo < 0 else base + lo, H, W, int(h_off), int(w_off))
                flags += [int(bool(flip)), int(k) & 3]
        else:
            with_label = tiles[order[0]][1] is not None
            for i, (h_off, w_off, flip, k) in zip(order, params):
                img, lbl = tiles[i]
                if lbl is not None and tuple(lbl.shape) != tuple(img.shape[:2]):
                    raise ValueError('tile labels must be uint8 [H,W]')
                # only the rows the crop reads cross the bus: a tile at least as tall as the crop is never padded vertically (base_dataset.py:88-104 pads a tile
                # that is SMALLER than the crop), so rows [h_off, h_off + ch) -- a contiguous slice of the row-major tile -- are all the kernel looks at (1.5 instead
                # of 3 MB for a 512-row crop of a 1024 x 1024 RGB tile)
                if img.shape[0] > ch and 0 <= h_off <= img.shape[0] - ch:
                    img = img[h_off:h_off + ch]
                    lbl = None if lbl is None else lbl[h_off:h_off + ch]
                    h_off = 0
                it = (img.contiguous() if isinstance(img, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(img))).to(self.device, non_blocking=True)
                if it.dtype != torch.uint8 or it.dim() != 3 or it.shape[2] != 3:
                    raise ValueError('tile images must be uint8 [H,W,3]')
                lt = None
                if lbl is not None:
                    lt = (lbl.contiguous() if isinstance(lbl, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(lbl))).to(self.device, non_blocking=True)
                    if lt.dtype != torch.uint8 or tuple(lt.shape) != tuple(it.shape[:2]):
                        raise ValueError('tile labels must be uint8 [H,W]')
                keep += [it, lt]
                rec += struct.pack('<QQiiii', it.data_ptr(), 0 if lt is None else lt.data_ptr(), it.shape[0], it.shape[1], int(h_off), int(w_off))
                flags += [int(bool(flip)), int(k) & 3]
        table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(self.device)
        fl = torch.tensor(flags, dtype=torch.int32).to(self.device)
        out = torch.empty((B, 3, ch, cw), dtype=torch.float32, device=self.device)
        lab = torch.empty((B, ch, cw), dtype=torch.int64, device=self.device) if with_label else None
        _lib.check(_lib.lib().sl_augment_batch(_p(table), _p(fl), B, ch, cw, self.mean, self.std, self.ignore_label, _p(self.lut), _p(out), _p(lab), _s()), 'augment_batch')
        for t in keep:                                  # the raw tiles must outlive the kernel on this stream
            if t is not None:
                t.record_stream(torch.cuda.current_stream())
        return out, lab
