"""CPU oracle of the OpenEarthMap tile preparation: TEST INFRASTRUCTURE ONLY.

numpy restatement of dataset/base_dataset.py (crop :140-174, pad :88-104, random_flip :106-110, fixed_random_rotate :134-138, normalize
:29-34, totensor :36-43), of the label re-indexing in dataset/oem.py:113-133 and of the novel-tile rule of dataset/oem_ft.py:197, with the
random draws made explicit (`draw_train_params` consumes numpy's / random's generators in the reference's order).  Pinned by
tests/golden/make_golden.py::g17 against the imported reference (cv2.copyMakeBorder, the one OpenCV call on this path, is replaced there by
its numpy equivalent because OpenCV is not installed in this image; rasterio's read() is fed synthetic arrays).  The fine-tune pair reader
(dataset/oem_ft.py:72-124 update_base_list, :126-181 _get_supp_list, :189-220 _get_train_sample, :262-299 _filter_and_map_ids) is restated
at the end and pinned by ::g19 the same way.
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


# ---------------------------------------------------------------------------------------------- fine-tune pair reader (dataset/oem_ft.py)
def ft_tiles(n=12, H=96, W=80, block=16):
    """Synthetic stand-ins for OpenEarthMap tiles of golden G19 (pure functions of the tile index): ids, images uint8 [H,W,3], labels uint8 [H,W]
    in the dataset's own numbering (0 = unlabeled, 1..7 base, 8..11 novel).  Class 7 exists in tile 3 only (fewer files than shots: the
    'extend images with repeating' branch), tiles 0-4 hold base classes only, one tile is unlabeled in its upper part."""
    import torch
    from oracle import formula as fm
    ids, imgs, labs = [], {}, {}
    for k in range(n):
        id_ = 't%02d' % k
        hb, wb = (H + block - 1) // block, (W + block - 1) // block
        coarse = (fm.uniform01('g19/lab%d' % k, hb * wb) * (8 if k < 5 else 12)).floor().to(torch.uint8).reshape(hb, wb)
        lab = coarse.repeat_interleave(block, 0).repeat_interleave(block, 1)[:H, :W].numpy().copy()
        if k != 3:
            lab[lab == 7] = 1
        else:
            lab[:block, :block] = 7
        if k == 6:
            lab[:40] = 0
        img = (fm.uniform01('g19/img%d' % k, H * W * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(H, W, 3).numpy()
        ids.append(id_); imgs[id_] = img; labs[id_] = lab
    return ids, imgs, labs


def filter_and_map_ids(ids, read_label, base_classes, novel_classes, filter_intersection=False):
    """oem_ft.py:262-299: class -> list of tile ids holding it (in list order); with filter_intersection a tile counts for its base classes
    only when it holds no novel class at all."""
    from collections import defaultdict
    base_cls_to_ids, novel_cls_to_ids = defaultdict(list), defaultdict(list)
    for id_ in ids:
        mask = read_label(id_)
        label_class = np.unique(mask).tolist()
        if 0 in label_class:
            label_class.remove(0)
        valid_base = set(np.unique(mask).tolist()) & set(base_classes)
        valid_novel = set(np.unique(mask).tolist()) & set(novel_classes)
        if valid_base and (not filter_intersection or set(label_class).issubset(set(base_classes))):
            for cls in valid_base:
                base_cls_to_ids[cls].append(id_)
        for cls in valid_novel:
            novel_cls_to_ids[cls].append(id_)
    return base_cls_to_ids, novel_cls_to_ids


def sample_base_ids(base_cls_to_ids, base_classes, shot):
    """oem_ft.py:72-124 == :126-181 (update_base_list and _get_supp_list draw identically): per base class `shot` tile ids -- every file once plus
    random.randint(1, n) - 1 repeats when the class has fewer files than shots, random.choices(range(n), k=shot) otherwise."""
    out = []
    for cls in list(base_classes):
        files = base_cls_to_ids[cls]
        n = len(files)
        if n < shot:
            out += [files[i] for i in range(n)]
            out += [files[random.randint(1, n) - 1] for _ in range(shot - n)]
        else:
            out += [files[j] for j in random.choices(list(range(n)), k=shot)]
    return out


def ft_pair(index, base_id_list, novel_id_list, read_image, read_label, crop_size, ignore_label=255):
    """oem_ft.py:189-220: random novel support tile (0 -> ignore BEFORE the crop draw, :197), the index-th base tile, two independent
    crop / pad / flip / rot90 draw sets (novel tile first).  -> (img, lbl, img_b, lbl_b, id, prm, prm_b)"""
    id_b = base_id_list[index]
    id_ = random.choice(novel_id_list)
    image, label = read_image(id_), read_label(id_)
    label = np.where(label == 0, ignore_label, label).astype(np.uint8)
    image_b, label_b = read_image(id_b), read_label(id_b)
    prm = draw_train_params(label, crop_size, ignore_label)
    img, lbl = prepare_tile(image, label, crop_size, *prm, ignore_label=ignore_label)
    prm_b = draw_train_params(label_b, crop_size, ignore_label)
    img_b, lbl_b = prepare_tile(image_b, label_b, crop_size, *prm_b, ignore_label=ignore_label)
    return img, lbl, img_b, lbl_b, id_, prm, prm_b
