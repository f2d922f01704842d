"""Shared pieces of the two training entry points (train_base.py / ft_pop.py of the reference): the command-line
surface, LR schedule, validation loop and checkpoint writer.  The flag names, defaults and meanings are the reference's
(train_base.py:47-111, ft_pop.py:47-115) so scripts/*.sh work with only the module path edited."""
import argparse
import os

import numpy as np
import torch

from .utils import pyt_utils as my_utils


def str2bool(v):
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


# (flag, kwargs) -- common to both drivers
_COMMON = [
    ('--dataset', dict(type=str, default='cityscapes', help='Dataset for training')),
    ('--batch-size', dict(type=int, default=8, help='Number of images sent to the network in one step.')),
    ('--data-dir', dict(type=str, default='/data/pascal-context', help='Path to the dataset directory.')),
    ('--train-list', dict(type=str, default='./dataset/list/context/train.txt', help='File listing the training images.')),
    ('--base-size', dict(type=str, default='1024,1024', help='Base size of images for resize.')),
    ('--input-size', dict(type=str, default='512,512', help='Comma-separated height,width of the training crops.')),
    ('--learning-rate', dict(type=float, default=1e-2, help='Base learning rate (polynomial decay).')),
    ('--momentum', dict(type=float, default=0.9, help='Momentum component of the optimiser.')),
    ('--power', dict(type=float, default=0.9, help='Decay parameter of the learning rate.')),
    ('--weight-decay', dict(type=float, default=0.0005, help='Regularisation parameter for L2-loss.')),
    ('--start-epoch', dict(type=int, default=0, help='Which epoch to start.')),
    ('--num-epoch', dict(type=int, default=100, help='Number of training epochs.')),
    ('--restore-from', dict(type=str, default='/home/model/resnet_backbone/resnet101-imagenet.pth', help='Where to restore model parameters from.')),
    ('--snapshot-dir', dict(type=str, default='/home/output', help='Where to save snapshots of the model.')),
    ('--model', dict(type=str, default='None', help='choose model.')),
    ('--num-workers', dict(type=int, default=4, help='choose the number of workers.')),
    ('--backbone', dict(type=str, default='resnet50', help='backbone model: resnet101, resnet50 (default)')),
    ('--os', dict(type=int, default=8, help='output stride')),
    ('--print-frequency', dict(type=int, default=100, help='Number of training steps between log lines.')),
    ('--save-pred-every', dict(type=int, default=5, help='Save summaries and checkpoint every often.')),
    ('--shot', dict(type=int, default=1, help='number of support pairs')),
    ('--val-list', dict(type=str, default='./dataset/list/context/val.txt', help='File listing the validation images.')),
    ('--test-batch-size', dict(type=int, default=1, help='Number of images sent to the network in one validation step.')),
    ('--filter-novel', dict(action='store_true', default=False, help='filter images containing novel classes during training.')),
    ('--freeze-backbone', dict(action='store_true', default=False, help='freeze the backbone during training.')),
    ('--no-step-graph', dict(action='store_true', default=False, help='issue every kernel of the training step from Python instead of replaying '
                                 'the step as one captured HIP graph (segland_amd/graph_step.py; single-GPU AdamW runs only).')),
    ('--allow-random-init', dict(action='store_true', default=False, help='continue with random weights when --restore-from does not exist '
                                                                           '(the reference fails in torch.load; so does this build without the flag).')),
    ('--fp16', dict(action='store_true', default=False, help='mixed precision: bf16 MFMA with fp32 accumulate on MI355X '
                                                                '(the reference uses fp16 autocast + GradScaler); default is exact-fp32 MFMA.')),
]
_BASE_ONLY = [
    ('--random-seed', dict(type=int, default=321, help='Random seed to have reproducible results.')),
    ('--fold', dict(type=int, default=0, choices=[-1, 0, 1, 2, 3], help='validation fold')),
    ('--fix-bn', dict(action='store_true', default=False, help='whether to fix batchnorm during training.')),
    ('--finetune', dict(action='store_true', default=False, help='whether to finetune the decoder.')),
    ('--single-step', dict(action='store_true', default=False, help='ONE AdamW step per iteration; default mirrors the reference, '
                                                                      'whose scaler.step + optimizer.step is two (train_base.py:262-264).')),
]
_FT_ONLY = [
    ('--random-seed', dict(type=str, default='123,234', help='Comma-separated seeds; one fine-tune run per seed.')),
    ('--fold', dict(type=int, default=0, choices=[0, 1, 2, 3], help='validation fold')),
    ('--fix-bn', dict(action='store_true', default=True, help='whether to fix batchnorm during training.')),
    ('--update-base', dict(action='store_true', default=False, help='whether to update base class with novel class.')),
    ('--update-epoch', dict(type=int, default=1, help='epoch interval for base-list update / validation.')),
    ('--fix-lr', dict(action='store_true', default=False, help='whether to fix learning rate during training.')),
]


def build_parser(ft=False):
    p = argparse.ArgumentParser(description='Few-shot Segmentation Framework (MI355X build)')
    for flag, kw in _COMMON + (_FT_ONLY if ft else _BASE_ONLY):
        p.add_argument(flag, **kw)
    return p


def lr_poly(base_lr, it, max_iter, power):
    return base_lr * ((1 - float(it) / max_iter) ** power)


def adjust_learning_rate_poly(optimizer, learning_rate, i_iter, max_iter, power, split=0, scale_lr=10.0):
    """Groups with index <= split get lr, the others lr*scale (train_base.py:116-128, ft_pop.py:119-131)."""
    lr = lr_poly(learning_rate, i_iter, max_iter, power)
    for index, group in enumerate(optimizer.param_groups):
        group['lr'] = lr if index <= split else lr * scale_lr
    return lr


def compute_dtype(args):
    return torch.bfloat16 if args.fp16 else torch.float32


_AUGMENTERS = None


def _augmenters_of(dataset):
    global _AUGMENTERS
    if _AUGMENTERS is None:
        import weakref
        _AUGMENTERS = weakref.WeakKeyDictionary()
    d = _AUGMENTERS.get(dataset)
    if d is None:
        d = _AUGMENTERS[dataset] = {}
    return d


def batch_to_device(batch, dataset, device):
    """(img, mask) on the device from a loader batch: ready tensors (synthetic), or raw uint8 tiles + the reference's random draws that the GPU
    crops / pads / flips / rotates / normalises / re-indexes in one launch (dataset/augment.py).  The augmenters (device-side lookup
    tables, ctypes staging) are cached per LIVE dataset object (_AUGMENTERS, weak keys): a cache keyed by id(dataset) would hand a later dataset that happens
    to get the same id the augmenter of a dead one (other crop size, statistics, label map); stored on the dataset itself they would be pickled into every
    DataLoader worker (the loaders are re-created every epoch; ctypes arrays do not pickle)."""
    if not getattr(dataset, 'raw_tiles', False):
        return batch[0].to(device, non_blocking=True), batch[1].to(device, non_blocking=True)
    tiles, params, _ = batch
    cache = _augmenters_of(dataset)
    if hasattr(dataset, 'crop_size'):
        key, make = ('train', str(device)), lambda: dataset.augmenter(device)
    else:                                  # validation: whole tiles, one augmenter per tile size
        size = tuple(tiles[0][0].shape[:2])
        key, make = (size, str(device)), lambda: dataset.augmenter(device, size)
    aug = cache.get(key)
    if aug is None:
        aug = cache[key] = make()
    return aug.prepare(tiles, params)


def ft_batch_to_device(batch, dataset, device):
    """(img, mask, img_b, mask_b) on the device from a fine-tune loader batch: ready tensors (synthetic_ft), or raw (novel, base) tile pairs +
    their draws (dataset/oem_ft.py, synthetic_raw_ft.py), whose 2B tiles are prepared by ONE GPU launch."""
    if not getattr(dataset, 'pair_tiles', False):
        return tuple(t.to(device, non_blocking=True) for t in batch[:4])
    cache = _augmenters_of(dataset)
    aug = cache.get(('pairs', str(device)))
    if aug is None:
        aug = cache[('pairs', str(device))] = dataset.augmenter(device)
    return aug.prepare(batch[0], batch[1])


def validate(model, dataloader, num_classes, ignore_label, device):
    """train_base.py:316-340 / ft_pop.py:312-336: logits -> upsample(align_corners=True) -> argmax -> IoU histogram,
    with the upsample+argmax fused in one HIP kernel (no H x W logits) and the histogram in another."""
    from . import ops
    model.eval()
    inter = torch.zeros(num_classes, device=device)
    union = torch.zeros(num_classes, device=device)
    for batch in dataloader:
        img, mask = batch_to_device(batch, dataloader.dataset, device)
        if mask is None:
            raise RuntimeError('validate(): the loader delivers unlabeled tiles; scoring needs labels (eval_base --save-prob dumps predictions for unlabeled tiles)')
        with torch.no_grad():
            logits = model(img)
            pred = ops.upsample_argmax(logits.float().contiguous(), mask.shape[1:])
        i, u, _ = my_utils.intersectionAndUnionGPU(pred, mask, num_classes, ignore_label)
        inter += i
        union += u
    return inter, union


def save_checkpoint(model, path):
    """Legacy (non-zipfile) serialisation and `module.`-prefixed keys, exactly the reference's on-disk format."""
    torch.save(model.state_dict(), path, _use_new_zipfile_serialization=False)


def save_training_state(model, optimizer, path, epoch, best=0.0, best_epoch=0):
    """Everything a true resume needs (the reference saves the bare model state_dict only, train_base.py:286-292, and parses
    `-c/--continue` without using it, engine.py:62-65): model weights in the reference's `module.`-prefixed format under 'state_dict' (so
    utils.pyt_utils.load_model reads this file too), the optimizer state (torch.optim layout), the finished epoch and the best validation score."""
    torch.save({'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': int(epoch), 'best': float(best), 'best_epoch': int(best_epoch)},
               path, _use_new_zipfile_serialization=False)


def load_training_state(model, optimizer, path):
    """-> (epoch to continue with, best, best_epoch).  `model` is the wrapped model (its keys carry `module.`)."""
    ckpt = torch.load(path, map_location='cpu')
    if 'optimizer' not in ckpt or 'state_dict' not in ckpt:
        raise RuntimeError('%s is not a training state written by save_training_state (a bare state_dict restores weights only: use --restore-from)' % path)
    model.load_state_dict(ckpt['state_dict'], strict=True)
    optimizer.load_state_dict(ckpt['optimizer'])
    from .functional import weights_changed
    weights_changed()                      # the derived GEMM-layout weight copies follow the loaded parameters
    return int(ckpt['epoch']), float(ckpt.get('best', 0.0)), int(ckpt.get('best_epoch', 0))


def miou(inter, union):
    return np.nanmean((inter / union).cpu().numpy())


def checkpoint_or_none(path, allow_random_init, what='--restore-from'):
    """The path if it exists; None (random initialisation) only when explicitly allowed.  The reference raises inside torch.load for a
    missing file (utils/pyt_utils.py:91); silently training / evaluating a random model is never what the caller meant."""
    import os.path as osp
    if path and osp.exists(str(path)):
        return path
    if allow_random_init:
        import logging
        logging.getLogger('Segmentation').error('%s=%r does not exist: continuing with RANDOM weights (--allow-random-init)', what, path)
        return None
    raise FileNotFoundError('%s=%r does not exist (pass --allow-random-init to run with random weights)' % (what, path))


def resolve(dataset_pkg, name):
    mod = getattr(dataset_pkg, name, None)
    if mod is None:
        have = sorted(k for k, v in vars(dataset_pkg).items() if hasattr(v, 'GFSSegTrain'))
        raise RuntimeError("unknown dataset '%s' (available: %s; 'oem' / 'oem_ft' decode GeoTIFF tiles with rasterio or Pillow)" % (name, ', '.join(have)))
    return mod
