#!/usr/bin/env python3
"""Feed rate of the tile preparation path: DataLoader workers produce RAW uint8 tiles + the reference's random draws
(dataset/synthetic_raw.py / synthetic_raw_ft.py: the sample format of the OpenEarthMap readers, decode replaced by a synthetic generator of the same
size), raw_collate / pair_collate, host -> device copies and ONE sl_augment_batch launch per batch.  Reports tiles/s for
  (a) the GPU stage alone on resident raw tiles (H2D copy + kernel, no loader),
  (b) the whole feed with N workers (what a training loop would see with a free GPU),
so that the figure can be read against the 600 tiles/s the training step consumes.  1024 x 1024 raw tiles, 512 x 512 crops (scripts/train_oem.sh).

    python tools/feed_rate.py [--workers 8] [--batch 16] [--batches 40] [--pairs]
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
    a = p.parse_args()
    dev = torch.device('cuda', 0)
    if a.pairs:
        from segland_amd.dataset import synthetic_raw_ft as ds_mod
        from segland_amd.dataset.oem_ft import pair_collate as collate
        ds = ds_mod.GFSSegTrain(shot=5, crop_size=(a.crop, a.crop), tile=(a.tile, a.tile), length=40)
        per_sample = 2
    else:
        from segland_amd.dataset import synthetic_raw as ds_mod
        from segland_amd.dataset.oem import raw_collate as collate
        ds = ds_mod.GFSSegTrain(crop_size=(a.crop, a.crop), length=a.batch * a.batches)
        ds.tile = (a.tile, a.tile)
        per_sample = 1
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
    print('feed_rate: %s, raw %dx%d -> crop %dx%d, batch %d: GPU stage alone (H2D + sl_augment_batch) %.0f tiles/s; whole feed with %d workers on %d host cores %.0f tiles/s'
          % ('fine-tune pairs' if a.pairs else 'base tiles', a.tile, a.tile, a.crop, a.crop, a.batch, gpu_rate, a.workers, cpus, feed_rate))


if __name__ == '__main__':
    main()
