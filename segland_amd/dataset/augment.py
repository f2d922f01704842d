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
    tiles: list of (image uint8 [H,W,3], label uint8 [H,W] or None) numpy arrays / CPU tensors; params: list of (h_off, w_off, flip, k)."""

    def __init__(self, crop_size, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), ignore_label=255, lut=None, device='cuda'):
        self.crop_size, self.ignore_label, self.device = tuple(crop_size), ignore_label, torch.device(device)
        self.mean, self.std = (C.c_double * 3)(*mean), (C.c_double * 3)(*std)      # float64 like the python lists numpy broadcasts in normalize()
        self.lut = None if lut is None else torch.as_tensor(np.asarray(lut, dtype=np.uint8)).to(self.device)

    def prepare(self, tiles, params):
        ch, cw = self.crop_size
        if any(k % 2 for _, _, _, k in params) and ch != cw:
            raise ValueError('rot90 by an odd count needs square crops')
        B = len(tiles)
        keep, rec, flags = [], b'', []
        with_label = tiles[0][1] is not None
        for (img, lbl), (h_off, w_off, flip, k) in zip(tiles, params):
            it = torch.as_tensor(np.ascontiguousarray(img)).to(self.device, non_blocking=True)
            if it.dtype != torch.uint8 or it.dim() != 3 or it.shape[2] != 3:
                raise ValueError('tile images must be uint8 [H,W,3]')
            lt = None
            if lbl is not None:
                lt = torch.as_tensor(np.ascontiguousarray(lbl)).to(self.device, non_blocking=True)
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
