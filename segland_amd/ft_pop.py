#!/usr/bin/env python3
"""Novel-class fine-tuning entry point (counterpart of the reference's ft_pop.py): frozen backbone/decoder/classifier,
trainable novel prototypes + classifier_n, pseudo-labelled base tiles, one run per seed."""
import os
import os.path as osp

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import dataset as dataset_pkg
from . import networks
from . import graph_step
from .drivers import adjust_learning_rate_poly, build_parser, checkpoint_or_none, compute_dtype, ft_batch_to_device, resolve, save_checkpoint, validate
from .engine import Engine
from .loss import get_loss
from .utils import pyt_utils as my_utils


def ft_iteration(model, optimizer, loss_scaler, batch, device, dataset=None):
    """Loop body of ft_pop.py:243-256 (single optimizer step, zero_grad afterwards)."""
    img, mask, img_b, mask_b = ft_batch_to_device(batch, dataset, device)
    loss_dict = model(img, mask, img_b, mask_b.contiguous())
    grad_norm = loss_scaler(loss_dict['total_loss'], optimizer, clip_grad=5.0, parameters=model.parameters())
    optimizer.zero_grad()
    return loss_dict, grad_norm


_CLIP_IN_STEP = True      # test hook: the clip coefficient enters the SGD launch; False: torch's clip_grad_norm_ (scales the gradients in place) + a plain step (profiles/r5_ab_ft_small_launches.txt)


def ft_graph_body(model, clip_grad=5.0, optimizer=None):
    """forward + backward + clip_grad_norm_ (+ the optimizer step) of ft_iteration as a function of the four batch tensors, for graph_step.GraphedStep.
    optimizer: a segland_amd.optim.SGD -- its one-launch step reads the learning rate from device memory, so it sits inside the graph although ft_pop changes the
    learning rate every iteration; with torch.optim.SGD (which bakes lr into its four launches) pass None and step behind the replay."""
    def body(img, mask, img_b, mask_b):
        loss_dict = model(img, mask, img_b, mask_b)
        loss_dict['total_loss'].backward()
        params = [p for p in model.parameters() if p.grad is not None]
        if optimizer is not None and _CLIP_IN_STEP:
            # the clip coefficient goes into the optimizer's one launch (as train_base.py's AdamW step does): the in-place scaling launches of clip_grad_norm_ fall away;
            # the gradients themselves stay unscaled (ft_iteration_graphed zeroes them right behind the step)
            from .optim import clip_coefficient
            norm, coef = clip_coefficient(params, clip_grad)
            optimizer.step(grad_scale=coef)
            return loss_dict, norm
        norm = torch.nn.utils.clip_grad_norm_(params, clip_grad)
        if optimizer is not None:
            optimizer.step()
        return loss_dict, norm
    return body


def ft_iteration_graphed(graphed, optimizer, batch, device, dataset=None):
    """ft_iteration with the model part replayed from one HIP graph."""
    img, mask, img_b, mask_b = ft_batch_to_device(batch, dataset, device)
    loss_dict, grad_norm = graphed(img, mask, img_b, mask_b.contiguous())
    if graphed.optimizer is None:                                 # torch.optim.SGD: not part of the body (with segland_amd.optim.SGD the body has stepped, eagerly or in the replay)
        optimizer.step()
    optimizer.zero_grad()
    return loss_dict, grad_norm


def main(argv=None):
    parser = build_parser(ft=True)
    with Engine(custom_parser=parser, argv=argv) as engine:
        args = engine.args
        logger = my_utils.prep_experiment(args) if engine.is_main else None
        input_size = tuple(map(int, args.input_size.split(',')))
        base_size = tuple(map(int, args.base_size.split(',')))
        for seed in map(int, args.random_seed.split(',')):
            my_utils.set_seed(seed)
            ds_ft, ds = resolve(dataset_pkg, args.dataset + '_ft'), resolve(dataset_pkg, args.dataset)
            trainset = ds_ft.GFSSegTrain(args.data_dir, args.train_list, args.fold, args.shot, crop_size=input_size, base_size=base_size,
                                         mode='train', seed=seed, filter=args.filter_novel)
            train_loader, train_sampler = engine.get_train_loader(trainset)
            args.ignore_label, args.base_classes = trainset.ignore_label, len(trainset.base_classes)
            testset = ds.GFSSegVal(args.data_dir, args.val_list, args.fold, base_size=base_size, resize_label=True, use_novel=True, use_base=True)
            test_loader, test_sampler = engine.get_test_loader(testset)
            args.novel_classes, args.num_classes = len(testset.novel_classes), testset.num_classes + 1
            if engine.distributed:
                test_sampler.set_epoch(0)
            criterion = get_loss(args)
            norm = nn.SyncBatchNorm if engine.distributed else nn.BatchNorm2d
            seg_model = getattr(networks, args.model).GFSS_Model(
                n_base=args.base_classes, criterion=criterion, backbone=args.backbone, norm_layer=norm, dilated=(args.os != 32), os=args.os,
                is_ft=True, n_novel=args.novel_classes, compute_dtype=compute_dtype(args))
            if checkpoint_or_none(args.restore_from, args.allow_random_init):
                my_utils.load_model(seg_model, args.restore_from, is_restore=True)
            seg_model.init_cls_n()
            params = my_utils.get_parameters(seg_model, lr=args.learning_rate, freeze_backbone=args.freeze_backbone)
            split = -1 if args.freeze_backbone else 0
            if engine.use_cuda:
                from .optim import SGD                                  # torch.optim.SGD semantics / state_dict, one capturable kernel launch per step
                optimizer = SGD(params, lr=args.learning_rate, momentum=args.momentum, weight_decay=args.weight_decay)
            else:
                optimizer = optim.SGD(params, lr=args.learning_rate, momentum=args.momentum, weight_decay=args.weight_decay)
            optimizer.zero_grad()
            model = engine.data_parallel(seg_model)
            loss_scaler = my_utils.NativeScalerWithGradNormCount()
            graphed = None
            if not args.no_step_graph and graph_step.eligible(model, optimizer, engine.device, need_adamw=False):
                in_graph = optimizer if hasattr(optimizer, 'graph_prepare') else None
                graphed = graph_step.GraphedStep(ft_graph_body(model, optimizer=in_graph), model, in_graph)
            if engine.is_main:
                os.makedirs(args.snapshot_dir, exist_ok=True)
            it, max_it = args.start_epoch * len(train_loader), args.num_epoch * len(train_loader)
            lr, best, best_b = args.learning_rate, 0.0, 0.0
            for epoch in range(args.start_epoch, args.num_epoch):
                my_utils.set_seed(seed + epoch)
                if engine.distributed:
                    train_sampler.set_epoch(epoch)
                model.module.train_mode()
                for i, batch in enumerate(train_loader):
                    it += 1
                    if not args.fix_lr:
                        lr = adjust_learning_rate_poly(optimizer, args.learning_rate, it - 1, max_it, args.power, split)   # per ITERATION here
                    if graphed is not None:
                        loss_dict, grad_norm = ft_iteration_graphed(graphed, optimizer, batch, engine.device, trainset)
                    else:
                        loss_dict, grad_norm = ft_iteration(model, optimizer, loss_scaler, batch, engine.device, trainset)
                    if i % args.print_frequency == 0:
                        vals = engine.reduce_loss_dict(loss_dict)
                        if engine.is_main:
                            logger.info('Epoch{}/Iters{} Iter{}/{}: lr={:.2e} grad_norm={:.4f}'.format(epoch + 1, it, i + 1, len(train_loader), lr, float(grad_norm))
                                        + ''.join(' %s=%.4f' % kv for kv in vals.items()))
                if args.update_base and (epoch + 1) % args.update_epoch == 0:
                    trainset.update_base_list()
                if epoch % args.update_epoch == 0 or epoch == args.num_epoch - 1:
                    inter, union = validate(model, test_loader, args.num_classes, args.ignore_label, engine.device)
                    inter, union = engine.all_reduce_tensor(inter, norm=False), engine.all_reduce_tensor(union, norm=False)
                    arr = (inter / union).cpu().numpy()
                    b_iou, n_iou, t_iou = np.nanmean(arr[:args.base_classes + 1]), np.nanmean(arr[args.base_classes + 1:]), np.nanmean(arr)
                    if engine.is_main:
                        if t_iou >= best and b_iou - best_b > 0.001:
                            save_checkpoint(model, osp.join(args.snapshot_dir, 'best_%d.pth' % seed))
                            best, best_b = t_iou, b_iou
                        logger.info('>>>>>>> Evaluation Results: meanIU: {:.2%}, baseIU: {:.2%}, novelIU: {:.2%}, best_IU: {:.2%} <<<<<<<'.format(t_iou, b_iou, n_iou, best))
                        if epoch % 50 == 0 or epoch == args.num_epoch - 1:
                            save_checkpoint(model, osp.join(args.snapshot_dir, 'epoch_%d_%d.pth' % (epoch, seed)))


if __name__ == '__main__':
    main()
