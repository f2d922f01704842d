"""Host-side helpers the drivers need (counterparts of the reference's utils/pyt_utils.py).
Only what the POP train / fine-tune / validate loops call is provided."""
import logging
import os
import random
import time
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist


def load_model(model, model_file, is_restore=False, backbone_only=False):
    """Load a checkpoint (path or state_dict) non-strictly.  `is_restore` strips the 7-character `module.` prefix that
    DataParallel/DDP checkpoints of the reference carry; `backbone_only` prefixes keys with `backbone.`
    (utils/pyt_utils.py:86-135)."""
    t0 = time.time()
    if isinstance(model_file, str):
        ckpt = torch.load(model_file, map_location='cpu')
        state = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
        if 'model' in state:
            state = state['model']
    else:
        state = model_file
    if is_restore:
        state = OrderedDict((k[7:], v) for k, v in state.items())
    if backbone_only:
        state = OrderedDict(('backbone.' + k, v) for k, v in state.items())
    model.load_state_dict(state, strict=False)
    own, got = set(model.state_dict().keys()), set(state.keys())
    if own - got:
        logging.warning('Missing key(s) in state_dict: %s', ', '.join(sorted(own - got)))
    if got - own:
        logging.warning('Unexpected key(s) in state_dict: %s', ', '.join(sorted(got - own)))
    logging.info('Load model from %s in %.2fs', model_file if isinstance(model_file, str) else '<state_dict>', time.time() - t0)
    return model


def get_parameters(model, lr, scale=10.0, freeze_backbone=False, fix_bn=False, logger=None):
    """Three AdamW/SGD groups keyed on substrings of the parameter name (utils/pyt_utils.py:216-249):
    backbone -> lr; non-backbone '*bias*' -> lr*scale, weight_decay 0; other non-backbone -> lr*scale."""
    wd_0, lr_1, lr_10 = [], [], []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        if 'backbone' not in key:
            (wd_0 if 'bias' in key else lr_10).append(value)
        elif freeze_backbone:
            value.requires_grad = False
        elif fix_bn and 'bn' in key:
            value.requires_grad = False
        else:
            lr_1.append(value)
    groups = [{'params': wd_0, 'lr': lr * scale, 'weight_decay': 0.0}, {'params': lr_10, 'lr': lr * scale}]
    if not freeze_backbone:
        groups.insert(0, {'params': lr_1, 'lr': lr})
    return groups


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def all_reduce_tensor(tensor, op=dist.ReduceOp.SUM, world_size=1, norm=True):
    with torch.no_grad():
        tensor = tensor.detach()
        dist.all_reduce(tensor, op)
        if norm:
            tensor.div_(world_size)
    return tensor


def intersectionAndUnionGPU(output, target, K, ignore_index=255):
    """utils/pyt_utils.py:293-305 on the HIP histogram kernel.  output/target: integer label maps of equal shape.
    Returns float tensors (area_intersection, area_union, area_target) of length K, like the reference."""
    from .. import ops
    assert output.shape == target.shape and output.dim() in (1, 2, 3)
    h = ops.iou_hist(output.reshape(-1).to(torch.uint8).contiguous(), target.reshape(-1).contiguous(), K, ignore_index).float()
    return h[0], h[1] + h[2] - h[0], h[2]


class NativeScalerWithGradNormCount:
    """Loss-scaler facade with the call signature of the reference (utils/pyt_utils.py:327-347).  The MI355X path
    computes in bf16 (no loss scaling needed) or fp32, so scaling is the identity; the call still does
    backward -> clip_grad_norm_ -> optimizer.step() exactly like GradScaler.step would."""
    state_dict_key = 'amp_scaler'

    def __init__(self, grad_div=1):
        self.grad_div = grad_div          # gradients are SUMS over this many ranks (Engine.data_parallel(sum_gradients=True))

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        params = [p for p in parameters if p.grad is not None]
        if clip_grad is not None and (hasattr(optimizer, 'repeat_next') or getattr(optimizer, 'supports_grad_scale', False)):
            from ..optim import clip_coefficient           # segland_amd.optim.AdamW / SGD apply the clip coefficient inside their kernel
            norm, coef = clip_coefficient(params, clip_grad, self.grad_div)
            optimizer.step(grad_scale=coef)
            return norm
        if self.grad_div != 1:
            raise RuntimeError('summed DDP gradients need segland_amd.optim.AdamW (the 1 / world_size lives in its kernel)')
        if clip_grad is not None:
            norm = torch.nn.utils.clip_grad_norm_(params, clip_grad)
        else:
            norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad.detach()) for p in params])) if params else torch.tensor(0.)
        optimizer.step()
        return norm

    def state_dict(self):
        return {}

    def load_state_dict(self, state_dict):
        pass


def get_logger(prefix, output_dir, date_str):
    logger = logging.getLogger('Segmentation')
    fmt = logging.Formatter(fmt='%(asctime)s.%(msecs)03d %(message)s', datefmt='%m-%d %H:%M:%S')
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    console.setFormatter(fmt)
    logger.addHandler(console)
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if rank == 0:
        fh = logging.FileHandler(os.path.join(output_dir, prefix + '_' + date_str + '.log'), 'w')
        fh.setFormatter(fmt)
        logger.addHandler(fh)
        logger.setLevel(logging.INFO)
    else:
        logger.setLevel(logging.ERROR)
    return logger


def prep_experiment(args, need_writer=False):
    log_path = os.path.join(args.snapshot_dir, 'log')
    os.makedirs(log_path, exist_ok=True)
    args.log_path = log_path
    args.date_str = datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
    logger = get_logger('', log_path, args.date_str)
    with open(os.path.join(args.snapshot_dir, args.date_str + '.txt'), 'w') as f:
        f.write(str(args) + '\n\n')
    return logger
