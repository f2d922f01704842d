#!/usr/bin/env python3
"""Generalised few-shot evaluation entry points (counterparts of the reference's eval_base.py / eval_ft.py) on the MI355X HIP path.

    python -m segland_amd.eval_base --model pspnet_pop --backbone resnet50 --dataset synthetic --restore-from best.pth --save-path out
    python -m segland_amd.eval_ft   ... --random-seed 123,234 --restore-from novel.pth      # loads novel_<seed>.pth per seed

Per batch: logits (eval-mode forward, folded BN) -> upsample(align_corners=True) + argmax in ONE kernel (the H x W logits of
eval_base.py:168 are never written) -> confusion-matrix kernel on the label grid; mIoU over base / novel / all classes from the
accumulated matrix exactly as eval_base.py:193-199.  The unlabeled-test branch (GeoTIFF + .mat dumps, eval_base.py:178-191) needs
rasterio and is row f-2 of SURVEY.md section 8.
"""
import argparse
import os
import os.path as osp

import numpy as np
import torch

from . import dataset as dataset_pkg
from . import networks
from . import ops
from .drivers import batch_to_device, checkpoint_or_none, compute_dtype, resolve, str2bool
from .engine import Engine
from .utils import pyt_utils as my_utils


def get_parser():
    """Flag names and defaults of eval_base.py:36-72 (+ --fp16 selecting bf16 MFMA, as in the training drivers)."""
    p = argparse.ArgumentParser(description='SegLand evaluation on the MI355X HIP path')
    p.add_argument('--dataset', type=str, default='synthetic')
    p.add_argument('--data-dir', type=str, default='')
    p.add_argument('--train-list', type=str, default='')
    p.add_argument('--val-list', type=str, default='')
    p.add_argument('--test-batch-size', type=int, default=1)
    p.add_argument('--model', type=str, default='pspnet_pop')
    p.add_argument('--restore-from', type=str, default='')
    p.add_argument('--backbone', type=str, default='resnet101')
    p.add_argument('--base-size', type=str, default='512,512')
    p.add_argument('--num-workers', type=int, default=0)
    p.add_argument('--save', type=str2bool, default='False')
    p.add_argument('--os', type=int, default=8)
    p.add_argument('--save-path', type=str, default='')
    p.add_argument('--random-seed', type=str, default='123')
    p.add_argument('--fold', type=int, default=0, choices=[0, 1, 2, 3])
    p.add_argument('--shot', type=int, default=1)
    p.add_argument('--fp16', action='store_true')
    p.add_argument('--save-prob', action='store_true', help="dump the upsampled logits of every tile as <save-path>/prob_<seed>/<id>.mat ({'outputs': [1,K,H,W]}, "
                   'eval_base.py:189-190) for segland_amd.fusemat, plus the predicted label map as a palette PNG')
    p.add_argument('--allow-random-init', action='store_true', help='evaluate random weights when --restore-from does not exist')
    return p


def confusion_of_batch(logits, label, num_classes, ignore_label, pad_to_longside=False):
    """eval_base.py:166-177 (eval_ft.py:166-181 with pad_to_longside): prediction at the label grid and its confusion counts."""
    from . import ops
    h, w = label.shape[-2:]
    if pad_to_longside:                     # eval_ft.py: upsample to a square of the long side, labels padded with ignore
        side = max(h, w)
        pad = torch.full((label.shape[0], side, side), ignore_label, dtype=label.dtype, device=label.device)
        pad[:, :h, :w] = label
        label, size = pad, (side, side)
    else:
        size = (h, w)
    pred = ops.upsample_argmax(logits.float().contiguous(), size)
    return pred, ops.confusion_matrix(pred, label.contiguous(), num_classes, ignore_label)


def write_prediction_tiff(path, pred_hw, source_tif=None):
    """eval_base.py:180-188: the label map of one tile as a single-band uint8 TIFF with the class colormap.  With rasterio installed and the tile's source image at hand
    the file is the reference's GeoTIFF (the source's profile -- CRS, transform -- with driver GTiff, dtype uint8, count 1, nodata 0, + write_colormap); without rasterio
    (the build image) Pillow writes the same pixels and the same palette as a baseline palette TIFF, without georeferencing.  Returns 'rasterio' or 'PIL'."""
    from .fusemat import COLORMAP
    pred_hw = np.ascontiguousarray(pred_hw, dtype=np.uint8)
    try:
        import rasterio
    except ImportError:
        rasterio = None
    if rasterio is not None and source_tif is not None and osp.exists(source_tif):
        with rasterio.open(source_tif) as src:
            profile = src.profile.copy()
        profile.update(driver='GTiff', dtype='uint8', count=1, nodata=0)
        with rasterio.open(path, 'w', **profile) as f:
            f.write(pred_hw, 1)
            f.write_colormap(1, {i: tuple(int(v) for v in COLORMAP[i % len(COLORMAP)]) for i in range(256)})
        return 'rasterio'
    from PIL import Image
    img = Image.fromarray(pred_hw, 'P')
    img.putpalette(np.resize(COLORMAP, (256, 3)).astype(np.uint8).tobytes())
    img.save(path, format='TIFF')
    return 'PIL'


def dump_probabilities(logits, size, pred, ids, out_dir, data_dir=None):
    """eval_base.py:168,180-190: per tile `<id>.mat` = {'outputs': [1,K,H,W] logits upsampled with align_corners=True} (the input of fusemat.py), `<id>.tif` = the
    argmax as a colormapped single-band TIFF next to the .mat directory (write_prediction_tiff: the reference's GeoTIFF when rasterio and the source tile
    <data_dir>/image/<id>.tif exist), and `<id>.png`, the same as a palette image."""
    import scipy.io
    from . import ops
    from .fusemat import COLORMAP
    os.makedirs(out_dir, exist_ok=True)
    up = ops.upsample_logits(logits.float().contiguous(), tuple(size)).cpu().numpy()
    pred = pred.cpu().numpy()
    for b in range(up.shape[0]):
        name = str(ids[b].item() if hasattr(ids[b], 'item') else ids[b])
        scipy.io.savemat(osp.join(out_dir, name + '.mat'), {'outputs': up[b:b + 1]})
        try:
            os.makedirs(out_dir + '_tif', exist_ok=True)
            write_prediction_tiff(osp.join(out_dir + '_tif', name + '.tif'), pred[b][:size[0], :size[1]],
                                  osp.join(data_dir, 'image', name + '.tif') if data_dir else None)
        except ImportError:
            pass
        try:
            from PIL import Image
            img = Image.fromarray(pred[b][:size[0], :size[1]], 'P')
            img.putpalette(np.resize(COLORMAP, (256, 3)))
            os.makedirs(out_dir + '_png', exist_ok=True)           # next to, not inside, the .mat directory fusemat walks
            img.save(osp.join(out_dir + '_png', name + '.png'))
        except ImportError:
            pass


def miou_from_confusion(cm, n_base):
    """eval_base.py:193-199: per-class IoU = tp / (row + column - tp); base = classes 0..n_base, novel = the rest."""
    cm = np.asarray(cm, dtype=np.float64)
    pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = tp / (pos + res - tp)
    return iou, float(np.nanmean(iou[:n_base + 1])), float(np.nanmean(iou[n_base + 1:])) if len(iou) > n_base + 1 else float('nan'), float(np.nanmean(iou))


def main(argv=None, ft=False):
    parser = get_parser()
    with Engine(custom_parser=parser, argv=argv) as engine:
        args = engine.args
        logger = None
        if engine.is_main and args.save_path:
            os.makedirs(args.save_path, exist_ok=True)
            from datetime import datetime
            logger = my_utils.get_logger('', args.save_path, datetime.now().strftime('%Y_%m_%d_%H_%M_%S'))
        args.base_size = tuple(map(int, args.base_size.split(',')))
        ds = resolve(dataset_pkg, args.dataset)
        testset = ds.GFSSegVal(args.data_dir, args.val_list, args.fold, base_size=args.base_size, resize_label=False, use_novel=True, use_base=True)
        test_loader, test_sampler = engine.get_test_loader(testset)
        args.ignore_label = testset.ignore_label
        args.base_classes, args.novel_classes = len(testset.base_classes), len(testset.novel_classes)
        args.num_classes = testset.num_classes + 1                      # background counts as a class (eval_base.py:120)
        if engine.distributed:
            test_sampler.set_epoch(0)
        assert args.os in (8, 16, 32)
        model_cls = getattr(networks, args.model).GFSS_Model
        seg_model = model_cls(n_base=args.base_classes, backbone=args.backbone, dilated=(args.os != 32), os=args.os,
                              n_novel=args.novel_classes, is_ft=ft, compute_dtype=compute_dtype(args))
        model = engine.data_parallel(seg_model.to(engine.device))
        results = {}
        for seed in map(int, args.random_seed.split(',')):
            path = (args.restore_from[:-4] + '_%d.pth' % seed) if ft else args.restore_from       # eval_ft.py:154
            if checkpoint_or_none(path, args.allow_random_init):
                my_utils.load_model(model, path)
            model.eval()
            cm = torch.zeros((args.num_classes, args.num_classes), dtype=torch.int64, device=engine.device)
            for batch in test_loader:
                # ready tensors (synthetic) or raw uint8 tiles + draws prepared on the GPU (oem / synthetic_raw: Engine installs raw_collate)
                image, label = batch_to_device(batch, testset, engine.device)
                ids = batch[2]
                with torch.no_grad():
                    logits = model(image)
                if label is None:
                    # unlabeled test tiles (eval_base.py:178-191): nothing to score -- the prediction and the probability dump are the product
                    size = tuple(image.shape[-2:])
                    pred = ops.upsample_argmax(logits.float().contiguous(), size)
                else:
                    size = tuple(label.shape[-2:])
                    pred, cmb = confusion_of_batch(logits, label, args.num_classes, args.ignore_label, pad_to_longside=ft)
                    cm += cmb
                if args.save_prob and args.save_path:
                    dump_probabilities(logits, size, pred, ids, osp.join(args.save_path, 'prob_%d' % seed), data_dir=getattr(args, 'data_dir', None))
            if engine.distributed:
                cm = engine.all_reduce_tensor(cm, norm=False)
            cmn = cm.cpu().numpy().astype(np.float64)
            iou, base_miou, novel_miou, total_miou = miou_from_confusion(cmn, args.base_classes)
            results[seed] = (base_miou, novel_miou, total_miou)
            if engine.is_main:
                if args.save_path:
                    np.save(osp.join(args.save_path, 'cmatrix_%d.npy' % seed), cmn)
                msg = ['>>>>>>> Current Seed %d: <<<<<<<' % seed, 'meanIoU---base: mIoU %.4f.' % base_miou,
                       'meanIoU---novel: mIoU %.4f.' % novel_miou, 'meanIoU---total: mIoU %.4f.' % total_miou]
                for m in msg:
                    (logger.info if logger else print)(m)
        return results


if __name__ == '__main__':
    main()
