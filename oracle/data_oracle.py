"""CPU oracle of the OpenEarthMap tile preparation: TEST INFRASTRUCTURE ONLY.

numpy restatement of dataset/base_dataset.py (crop :140-174, pad :88-104, random_flip :106-110, fixed_random_rotate :134-138, normalize
:29-34, totensor :36-43), of the label re-indexing in dataset/oem.py:113-133 and of the novel-tile rule of dataset/oem_ft.py:197, with the
random draws made explicit (`draw_train_params` consumes numpy's / random's generators in the reference's order).  Pinned by
tests/golden/make_golden.py::g17 against the imported reference (cv2.copyMakeBorder, the one OpenCV call on this path, is replaced there by
its numpy equivalent because OpenCV is not installed in this image; rasterio's read() is fed synthetic arrays).
Only tests/ may import this module."""
import random

import numpy as np


def draw_train_params(label, crop_size, ignore_label=255):
    """(h_off, w_off, flip, k) as _get_train_sample draws them (oem.py:70-74): crop offsets from np.random (redrawn while the crop holds nothing
    but ignore, base_dataset.py:146-155), then random.random() < 0.5 for the flip and int(random.random() // 0.25) for the rot90 count."""
    H, W = label.shape
    ch, cw = crop_size
    mh, mw = max(H - ch, 0), max(W - cw, 0)
    while True:
        h_off, w_off = np.random.randint(0, mh + 1), np.random.randint(0, mw + 1)
        u = np.unique(label[h_off:h_off + ch, w_off:w_off + cw]).tolist()
        if not (len(u) == 1 and ignore_label in u):
            break
    flip = random.random() < 0.5
    k = int(random.random() // 0.25)
    return h_off, w_off, flip, k


def val_crop_offsets(H, W, crop_size):
    # base_dataset.py:171-172: centred crop outside train mode
    return int(round(max(H - crop_size[0], 0) / 2.)), int(round(max(W - crop_size[1], 0) / 2.))


def prepare_tile(image, label, crop_size, h_off, w_off, flip, k, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), ignore_label=255):
    """image [H,W,3] uint8, label [H,W] uint8 or None -> (float32 [3,ch,cw], int64 [ch,cw] or None)."""
    ch, cw = crop_size
    img = image[h_off:h_off + ch, w_off:w_off + cw]
    lab = None if label is None else label[h_off:h_off + ch, w_off:w_off + cw]
    ph, pw = max(ch - img.shape[0], 0), max(cw - img.shape[1], 0)
    if ph or pw:                                                            # cv2.copyMakeBorder(BORDER_CONSTANT): zeros / ignore
        img = np.pad(img, ((0, ph), (0, pw), (0, 0)), constant_values=0)
        if lab is not None:
            lab = np.pad(lab, ((0, ph), (0, pw)), constant_values=ignore_label)
    if flip:
        img = np.flip(img, axis=1)
        lab = None if lab is None else np.flip(lab, axis=1)
    img = np.rot90(img, k, (0, 1))
    lab = None if lab is None else np.rot90(lab, k, (0, 1))
    img = img.astype(np.float32)[:, :, ::-1]
    img = img / 255.0
    img -= np.asarray(mean, dtype=np.float64)      # the reference subtracts python lists: numpy broadcasts them as float64, result stays float32
    img /= np.asarray(std, dtype=np.float64)
    return np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32), (None if lab is None else np.ascontiguousarray(lab).astype(np.int64))


def remap_lut(base_classes, novel_classes, use_base=True, use_novel=True):
    """dataset/oem.py:113-133 as a 256-entry table: base class c -> its rank + 1, novel class c -> rank + len(base) + 1 (or + 1 without base),
    classes switched off -> 0, every other value (0, 255) unchanged."""
    lut = np.arange(256, dtype=np.uint8)
    base, novel = list(base_classes), list(novel_classes)
    for c in range(256):
        if c in base:
            lut[c] = base.index(c) + 1 if use_base else 0
        elif c in novel:
            lut[c] = (novel.index(c) + (len(base) + 1 if use_base else 1)) if use_novel else 0
    return lut


def novel_tile_lut(ignore_label=255):
    # oem_ft.py:197: label = np.where(label == 0, ignore_label, label) on the novel tile of a fine-tuning pair
    lut = np.arange(256, dtype=np.uint8)
    lut[0] = ignore_label
    return lut


def fuse_probability_maps(mats):
    """fusemat.py:35-52: `mats[idx] += prob` in list order (float32, in place), then argmax(mat / n, axis=0) -> uint8 [H,W]."""
    acc = np.array(mats[0], dtype=np.float32, copy=True)
    for m in mats[1:]:
        acc += np.asarray(m, dtype=np.float32)
    return np.argmax(acc / len(mats), axis=0).astype(np.uint8)
