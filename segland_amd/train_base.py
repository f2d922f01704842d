#!/usr/bin/env python3
"""Base-class training entry point (counterpart of the reference's train_base.py) on the MI355X HIP path.

    python -m segland_amd.train_base --model pspnet_pop --backbone resnet50 --dataset synthetic --batch-size 16 --fp16 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m segland_amd.train_base ...
"""
import os
import os.path as osp

import torch
import torch.nn as nn
import torch.optim as optim

from . import dataset as dataset_pkg
from . import networks
from . import bucket_step
from . import graph_step
from .drivers import adjust_learning_rate_poly, batch_to_device, build_parser, load_training_state, save_training_state, checkpoint_or_none, compute_dtype, miou, resolve, save_checkpoint, validate
from .engine import Engine
from .loss import get_loss
from .utils import pyt_utils as my_utils


def train_iteration(model, optimizer, loss_scaler, img, mask, double_step=True):
    """Loop body of train_base.py:250-264."""
    optimizer.zero_grad()
    loss_dict = model(img, mask)
    fold = double_step and hasattr(optimizer, 'repeat_next')
    if fold:
        optimizer.repeat_next = 2     # segland_amd.optim.AdamW: both steps of the reference in one pass over the optimizer state
    grad_norm = loss_scaler(loss_dict['total_loss'], optimizer, clip_grad=5.0, parameters=model.parameters())
    if double_step and not fold:
        optimizer.step()          # the reference's second step on the same gradients
    return loss_dict, grad_norm


def main(argv=None):
    parser = build_parser(ft=False)
    with Engine(custom_parser=parser, argv=argv) as engine:
        args = engine.args
        logger = my_utils.prep_experiment(args) if engine.is_main else None
        if args.random_seed > 0:
            my_utils.set_seed(args.random_seed)
        args.input_size = tuple(map(int, args.input_size.split(',')))
        args.base_size = tuple(map(int, args.base_size.split(',')))
        ds = resolve(dataset_pkg, args.dataset)
        trainset = ds.GFSSegTrain(args.data_dir, args.train_list, args.fold, args.shot, crop_size=args.input_size,
                                  base_size=args.base_size, mode='train', filter=args.filter_novel)
        train_loader, train_sampler = engine.get_train_loader(trainset)
        args.ignore_label, args.num_classes = trainset.ignore_label, trainset.num_classes
        args.base_classes = len(trainset.base_classes)
        testset = ds.GFSSegVal(args.data_dir, args.val_list, args.fold, base_size=args.base_size, resize_label=False, use_novel=False)
        test_loader, test_sampler = engine.get_test_loader(testset)
        if engine.distributed:
            test_sampler.set_epoch(0)

        criterion = get_loss(args)
        norm = nn.SyncBatchNorm if engine.distributed else nn.BatchNorm2d       # parameter holders; statistics are per GPU
        assert args.os in (8, 16, 32)
        model_cls = getattr(networks, args.model).GFSS_Model                    # `networks.<model>.GFSS_Model`
        seg_model = model_cls(n_base=args.base_classes, criterion=criterion, backbone=args.backbone, norm_layer=norm,
                              # a true resume (-c state_N.pth) restores every tensor: no pretrained backbone needed, and no existence check on --restore-from
                              pretrained_model=(checkpoint_or_none(args.restore_from, args.allow_random_init)
                                                if args.start_epoch == 0 and not engine.continue_state_object else None),
                              dilated=(args.os != 32), os=args.os, compute_dtype=compute_dtype(args))
        if args.freeze_backbone and not engine.continue_state_object and checkpoint_or_none(args.restore_from, args.allow_random_init):
            my_utils.load_model(seg_model, args.restore_from, backbone_only=args.finetune, is_restore=not args.finetune)
        params = my_utils.get_parameters(seg_model, lr=args.learning_rate, freeze_backbone=args.freeze_backbone)
        if engine.use_cuda:
            from .optim import AdamW                                    # torch.optim.AdamW semantics / state_dict, one kernel launch per step
            optimizer = AdamW(params, lr=args.learning_rate, weight_decay=args.weight_decay)
        else:
            optimizer = optim.AdamW(params, lr=args.learning_rate, weight_decay=args.weight_decay)
        model = engine.data_parallel(seg_model, sum_gradients=engine.use_cuda,      # our AdamW divides by the world size inside its kernel
                                     graphable=engine.use_cuda and not args.no_step_graph)
        loss_scaler = my_utils.NativeScalerWithGradNormCount(engine.grad_div)
        if engine.is_main:
            os.makedirs(args.snapshot_dir, exist_ok=True)

        step = train_iteration
        if isinstance(model, bucket_step.BucketedReplica):
            # N > 1 ranks: graph A (forward + backward into the gradient buckets), one all-reduce per bucket, graph B (clip + AdamW)
            graphed = bucket_step.GraphedBucketStep(model, optimizer, double_step=not args.single_step)
            step = lambda m, o, s, img, mask, double_step: graphed(img, mask)      # noqa: E731
        elif not args.no_step_graph and graph_step.eligible(model, optimizer, engine.device):
            graphed = graph_step.GraphedTrainStep(train_iteration, model, optimizer, loss_scaler, double_step=not args.single_step)
            step = lambda m, o, s, img, mask, double_step: graphed(img, mask)      # noqa: E731  (one HIP graph launch per step)

        best, best_epoch = 0.0, 0
        if engine.continue_state_object:                                # -c / --continue FILE (engine.py:62-65 parses it; the reference never uses it)
            args.start_epoch, best, best_epoch = load_training_state(model, optimizer, engine.continue_state_object)
            if engine.is_main:
                logger.info('continuing from %s: epoch %d, best mIoU %.4f', engine.continue_state_object, args.start_epoch, best)
        it = args.start_epoch * len(train_loader)
        for epoch in range(args.start_epoch, args.num_epoch):
            if args.random_seed > 0:
                my_utils.set_seed(args.random_seed + epoch)
            if engine.distributed:
                train_sampler.set_epoch(epoch)
            if args.freeze_backbone:
                # train_base.py:244 passes backbone_only=args.finetune; pspnet_pop.GFSS_Model.train_mode() takes no argument there (the
                # reference crashes), swin_pop's does (swin_pop.py:220)
                import inspect
                tm = model.module.train_mode
                tm(backbone_only=args.finetune) if 'backbone_only' in inspect.signature(tm).parameters else tm()
            else:
                model.train()
            lr = adjust_learning_rate_poly(optimizer, args.learning_rate, epoch, args.num_epoch, args.power,
                                           split=-1 if args.freeze_backbone else 0)      # per EPOCH (train_base.py:248)
            for i, batch in enumerate(train_loader):
                it += 1
                img, mask = batch_to_device(batch, trainset, engine.device)
                loss_dict, grad_norm = step(model, optimizer, loss_scaler, img, mask, double_step=not args.single_step)
                if i % args.print_frequency == 0:
                    vals = engine.reduce_loss_dict(loss_dict)
                    if engine.is_main:
                        logger.info('Epoch{}/Iters{} Iter{}/{}: lr={:.2e} grad_norm={:.4f}'.format(epoch + 1, it, i + 1, len(train_loader), lr, float(grad_norm))
                                    + ''.join(' %s=%.4f' % kv for kv in vals.items()))
            e1 = epoch + 1
            if engine.is_main and (e1 % 10 == 0 or e1 >= args.num_epoch):
                save_checkpoint(model, osp.join(args.snapshot_dir, 'epoch_%d.pth' % e1))
            if e1 > 35 and (e1 % 10 == 0 or epoch == args.num_epoch - 1):
                inter, union = validate(model, test_loader, args.base_classes + 1, args.ignore_label, engine.device)
                inter, union = engine.all_reduce_tensor(inter, norm=False), engine.all_reduce_tensor(union, norm=False)
                m = miou(inter, union)
                if engine.is_main:
                    if m >= best:
                        save_checkpoint(model, osp.join(args.snapshot_dir, 'best.pth'))
                        best, best_epoch = m, e1
                    logger.info('>>>>>>> Evaluation Results: meanIU: {:.2%}, best_IU: {:.2%}, best_epoch: {} <<<<<<<'.format(m, best, best_epoch))
            if engine.is_main and (e1 % 10 == 0 or e1 >= args.num_epoch):
                # AFTER this epoch's validation, so a resume from state_<e1>.pth carries the epoch's best / best_epoch
                save_training_state(model, optimizer, osp.join(args.snapshot_dir, 'state_%d.pth' % e1), e1, best, best_epoch)

if __name__ == '__main__':
    main()
