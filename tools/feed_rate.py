#!/usr/bin/env python3
"""Feed rate of the tile preparation path: DataLoader workers produce RAW uint8 tiles + the reference's random draws
(dataset/synthetic_raw.py / synthetic_raw_ft.py: the sample format of the OpenEarthMap readers, decode replaced by a synthetic generator of the same
size), raw_collate / pair_collate, host -> device copies and ONE sl_augment_batch launch per batch.  Reports tiles/s for
  (a) the GPU stage alone on resident raw tiles (H2D copy + kernel, no loader),
  (b) the whole feed with N workers (what a training loop would see with a free GPU),
so that the figure can be read against the 600 tiles/s the training step consumes.  1024 x 1024 raw tiles, 512 x 512 crops (scripts/train_oem.sh).

--source tiff (default): the REAL readers (dataset/oem.py, dataset/oem_ft.py) on a generated directory of TIFF tiles (dataset/synthetic_tiff.py): every sample is
parsed from a file by the product's decoder (dataset/tiff.py: rasterio, else Pillow / libtiff); --compression picks the file flavour.  --source randint: the
decode-free synthetic generator of rounds 2-3 (a ~30 ms torch.randint per tile).  Fine-tune pairs: an epoch of the reference's reader is 7 classes x `shot` pairs
(35 at the 5-shot setting, ft_oem.sh runs batch size 1); with --batch 16 that is TWO batches per epoch, each made by one worker -- the flat 121 pairs/s of round 3
whatever the worker count.  The steady state is measured with --shot large enough for many batches per epoch; --batch 1 is the driver's own setting.

    python tools/feed_rate.py [--workers 16] [--batch 16] [--batches 40] [--pairs] [--shot 40] [--source tiff|randint] [--compression tiff_lzw]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--workers', type=int, default=8)
    p.add_argument('--batch', type=int, default=16)
    p.add_argument('--batches', type=int, default=40)
    p.add_argument('--tile', type=int, default=1024)
    p.add_argument('--crop', type=int, default=512)
    p.add_argument('--pairs', action='store_true', help='fine-tune pairs (dataset/oem_ft.py format): 2 tiles per sample')
    p.add_argument('--shot', type=int, default=40, help='pairs: tiles per base class, i.e. 7 x shot pairs per epoch (the 5-shot setting has 35)')
    p.add_argument('--source', default='tiff', choices=['tiff', 'randint'])
    p.add_argument('--compression', default=None, help='tiff_lzw | tiff_adobe_deflate | packbits (default: uncompressed)')
    p.add_argument('--files', type=int, default=96, help='tiff: tiles written to disk')
    p.add_argument('--repeat', type=int, default=8, help='tiff, base tiles: how often train.txt lists each file (epoch = files x repeat samples: a 96-sample epoch is six batches of 16,\n                   each made by ONE worker -- the loader then measures epoch restarts, not the feed)')
    a = p.parse_args()
    dev = torch.device('cuda', 0)
    decode = 'torch.randint stand-in'
    if a.source == 'tiff':
        import tempfile
        from segland_amd.dataset import synthetic_tiff, tiff
        from segland_amd.dataset import oem, oem_ft
        t0 = time.perf_counter()
        root = synthetic_tiff.make_dataset(tempfile.mkdtemp(prefix='feed_tiff_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None), n=a.files, tile=(a.tile, a.tile),
                                           seed=123, shot=a.shot, compression=a.compression, n_val=1, repeat=1 if a.pairs else a.repeat)
        lst = os.path.join(root, 'list', 'train.txt')
        nbytes = sum(os.path.getsize(os.path.join(root, 'images', f)) for f in os.listdir(os.path.join(root, 'images')))
        t0 = time.perf_counter()
        for k in range(8):
            tiff.read_tiff(os.path.join(root, 'images', 't%04d.tif' % k)); tiff.read_tiff(os.path.join(root, 'labels', 't%04d.tif' % k))
        decode = '%s decode of %s TIFF files (%.1f MB per image file, %.1f ms per tile = image + label in one process)' % (
            tiff.backend(), a.compression or 'uncompressed', nbytes / (a.files + 1) / 1e6, (time.perf_counter() - t0) / 8 * 1e3)
        if a.pairs:
            ds = oem_ft.GFSSegTrain(root, lst, 0, shot=a.shot, crop_size=(a.crop, a.crop), seed=123)
            collate, per_sample = ds.collate_fn, 2
        else:
            ds = oem.GFSSegTrain(root, lst, 0, crop_size=(a.crop, a.crop))
            collate, per_sample = ds.collate_fn, 1
    elif a.pairs:
        from segland_amd.dataset import synthetic_raw_ft as ds_mod
        from segland_amd.dataset.oem_ft import pair_collate as collate
        ds = ds_mod.GFSSegTrain(shot=a.shot, crop_size=(a.crop, a.crop), tile=(a.tile, a.tile), length=40)
        collate, per_sample = ds.collate_fn, 2
    else:
        from segland_amd.dataset import synthetic_raw as ds_mod
        from segland_amd.dataset.oem import raw_collate as collate
        ds = ds_mod.GFSSegTrain(crop_size=(a.crop, a.crop), length=a.batch * a.batches)
        ds.tile = (a.tile, a.tile)
        collate, per_sample = ds.collate_fn, 1
    aug = ds.augmenter(dev)
    # (a) GPU stage alone: one batch of raw tiles, prepared repeatedly (includes the H2D copies of the raw uint8 tiles: they are part of the stage)
    batch = collate([ds[i % len(ds)] for i in range(a.batch)])
    for _ in range(3):
        aug.prepare(batch[0], batch[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        aug.prepare(batch[0], batch[1])
    torch.cuda.synchronize()
    gpu_rate = n * a.batch * per_sample / (time.perf_counter() - t0)
    # (b) the whole feed
    dl = torch.utils.data.DataLoader(ds, batch_size=a.batch, num_workers=a.workers, collate_fn=collate, shuffle=False, drop_last=True,
                                     persistent_workers=a.workers > 0, prefetch_factor=4 if a.workers > 0 else None)
    it = iter(dl)
    first = next(it)
    aug.prepare(first[0], first[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tiles = 0
    while tiles < a.batch * a.batches * per_sample:
        try:
            b = next(it)
        except StopIteration:
            it = iter(dl)
            b = next(it)
        aug.prepare(b[0], b[1])
        tiles += len(b[0]) * per_sample
    torch.cuda.synchronize()
    feed_rate = tiles / (time.perf_counter() - t0)
    cpus = len(os.sched_getaffinity(0))
    unit = 'pairs/s' if a.pairs else 'tiles/s'
    print('feed_rate: %s, raw %dx%d -> crop %dx%d, batch %d, %d samples per epoch, %s: GPU stage alone (H2D + sl_augment_batch) %.0f %s; whole feed with %d workers on %d host cores %.0f %s'
          % ('fine-tune pairs' if a.pairs else 'base tiles', a.tile, a.tile, a.crop, a.crop, a.batch, len(ds), decode, gpu_rate / per_sample, unit, a.workers, cpus, feed_rate / per_sample, unit))
    if a.source == 'tiff':
        import shutil
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
