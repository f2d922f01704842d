"""CPU restatement of SegLand's PSPNet-POP hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference file:line it restates (paths relative to the upstream
repo LiZhuoHong/SegLand).  The arithmetic is the same ATen CPU fp32 arithmetic the
reference executes (torch 2.10 semantics: bilinear both align modes, adaptive-avg-pool bin
rule, F.normalize eps 1e-12, BN eps 1e-5 / momentum 0.1 / unbiased running_var, CE mean
over valid pixels), written as plain functions over a parameter tree whose state_dict keys
equal the reference's, so the same formula weights load into both.

Parity: PINNED by tests/golden/*.npz (generated from the imported reference by
tests/golden/make_golden.py, which also asserts oracle == reference there).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
D_MODEL = 512  # networks/pspnet_pop.py:43
PPM_BINS = (1, 2, 3, 6)  # networks/pspnet_pop.py:13


# ----------------------------------------------------------------------------- parameter tree
def _conv(cin, cout, k, stride=1, pad=0, dil=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=bias)


class _Holder(nn.Module):
    """Parameter container without behaviour; behaviour lives in the functions below."""


def make_bottleneck(inplanes, planes, stride, dilation, with_ds, last_relu=True):
    # networks/backbones/resnet.py:42-55 (ctor) -- conv1 1x1, conv2 3x3 (stride, pad=dil), conv3 1x1 x4
    m = _Holder()
    m.conv1 = _conv(inplanes, planes, 1)
    m.bn1 = nn.BatchNorm2d(planes)
    m.conv2 = _conv(planes, planes, 3, stride=stride, pad=dilation, dil=dilation)
    m.bn2 = nn.BatchNorm2d(planes)
    m.conv3 = _conv(planes, planes * 4, 1)
    m.bn3 = nn.BatchNorm2d(planes * 4)
    if with_ds:
        # networks/backbones/resnet.py:107-111
        m.downsample = nn.Sequential(_conv(inplanes, planes * 4, 1, stride=stride), nn.BatchNorm2d(planes * 4))
    else:
        m.downsample = None
    m.last_relu = last_relu
    return m


def make_resnet(layers, dilated=True, os=8, multi_grid=False, relu_l3=True, relu_l4=True):
    # networks/backbones/resnet.py:81-121
    net = _Holder()
    net.conv1 = _conv(3, 64, 7, stride=2, pad=3)
    net.bn1 = nn.BatchNorm2d(64)
    state = {'inplanes': 64}

    def stage(planes, n, stride=1, dilation=1, grid=1, last_relu=True):
        # resnet.py:105-121: block i gets dilation * grid[i % len(grid)]; only the LAST block may drop its final ReLU
        mg = (lambda i: grid[i % len(grid)]) if isinstance(grid, tuple) else (lambda i: 1)
        blocks = []
        need_ds = stride != 1 or state['inplanes'] != planes * 4
        blocks.append(make_bottleneck(state['inplanes'], planes, stride, dilation * mg(0), need_ds))
        state['inplanes'] = planes * 4
        for i in range(1, n):
            blocks.append(make_bottleneck(state['inplanes'], planes, 1, dilation * mg(i), False, last_relu=True if i != n - 1 else last_relu))
        return nn.Sequential(*blocks)

    grid = (1, 2, 4) if multi_grid else (1, 1, 1)
    net.layer1 = stage(64, layers[0])
    net.layer2 = stage(128, layers[1], stride=2)
    if dilated and os == 8:      # resnet.py:95-97
        net.layer3 = stage(256, layers[2], stride=1, dilation=2, last_relu=relu_l3)
        net.layer4 = stage(512, layers[3], stride=1, dilation=4, grid=grid, last_relu=relu_l4)
    elif dilated:                # resnet.py:98-100
        net.layer3 = stage(256, layers[2], stride=2, last_relu=relu_l3)
        net.layer4 = stage(512, layers[3], stride=1, dilation=2, grid=grid, last_relu=relu_l4)
    else:                        # resnet.py:101-103
        net.layer3 = stage(256, layers[2], stride=2, last_relu=relu_l3)
        net.layer4 = stage(512, layers[3], stride=2, last_relu=relu_l4)
    return net


RESNET_LAYERS = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3)}  # backbones/__init__.py:9-14


def make_ppm(features, out_features, sizes=PPM_BINS):
    # networks/pspnet_pop.py:13-29
    m = _Holder()
    m.sizes = tuple(sizes)
    m.stages = nn.ModuleList([
        nn.Sequential(nn.Identity(), _conv(features, out_features, 1), nn.BatchNorm2d(out_features), nn.Identity())
        for _ in sizes])
    m.bottleneck = nn.Sequential(
        _conv(features + len(sizes) * out_features, out_features, 3, pad=1),
        nn.BatchNorm2d(out_features), nn.Identity(), _conv(out_features, out_features, 1, bias=True))
    return m


def make_classifier(d):
    # networks/pspnet_pop.py:46-52
    return nn.Sequential(_conv(d, d, 1), nn.Identity(), _conv(d, d, 1), nn.Identity(), _conv(d, 1, 1))


class PopOracle(nn.Module):
    """Same constructor surface / state_dict keys as networks/pspnet_pop.py:37-74."""

    def __init__(self, n_base, criterion=None, is_ft=False, n_novel=0, backbone='resnet50',
                 dilated=True, os=8, d_model=D_MODEL, feat_channels=2048, _custom_backbone=None, multi_grid=False, relu_l3=True, relu_l4=True):
        super().__init__()
        if is_ft:   # pspnet_pop.py:54-56 -- own parameters are registered before child modules
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=False)
            self.novel_emb = nn.Parameter(torch.zeros(n_novel, d_model), requires_grad=True)
        else:       # pspnet_pop.py:67-69
            self.base_emb = nn.Parameter(torch.zeros(n_base, d_model), requires_grad=True)
            self.novel_emb = None
        self.backbone = _custom_backbone if _custom_backbone is not None else make_resnet(RESNET_LAYERS[backbone], dilated, os, multi_grid, relu_l3, relu_l4)
        self.decoder = make_ppm(feat_channels, d_model)
        self.classifier = make_classifier(d_model)
        if is_ft:
            self.classifier_n = make_classifier(d_model)
            nn.init.orthogonal_(self.novel_emb)
            ft_freeze(self)
        else:
            nn.init.orthogonal_(self.base_emb)
        self.n_base, self.n_novel, self.is_ft, self.criterion = n_base, n_novel, is_ft, criterion

    def forward(self, img, mask=None, img_b=None, mask_b=None):
        # dispatch of pspnet_pop.py:123-134
        if self.is_ft:
            if self.training:
                return forward_novel(self, img, mask, img_b, mask_b)
            return forward_all(self, img)
        return forward_base(self, img, mask)


def ft_freeze(model):
    # pspnet_pop.py:87-93
    for part in (model.backbone, model.decoder, model.classifier):
        for p in part.parameters():
            p.requires_grad = False


def train_mode(model):
    # pspnet_pop.py:80-84: everything train, backbone+decoder eval (BN uses running stats)
    model.train()
    model.backbone.eval()
    model.decoder.eval()


def init_cls_n(model):
    # pspnet_pop.py:76-78
    for src, dst in zip(model.classifier.parameters(), model.classifier_n.parameters()):
        dst.data.copy_(src.data)


# ----------------------------------------------------------------------------- forward pieces
def _bn(x, bn):
    # nn.BatchNorm2d.forward: batch statistics + running update in training, running stats in eval
    if bn.training:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, BN_MOMENTUM, BN_EPS)


def _cv(x, c):
    return F.conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation)


def bottleneck_forward(blk, x):
    # networks/backbones/resnet.py:57-78
    out = F.relu(_bn(_cv(x, blk.conv1), blk.bn1))
    out = F.relu(_bn(_cv(out, blk.conv2), blk.bn2))
    out = _bn(_cv(out, blk.conv3), blk.bn3)
    res = x if blk.downsample is None else _bn(_cv(x, blk.downsample[0]), blk.downsample[1])
    out = out + res
    return F.relu(out) if blk.last_relu else out


def resnet_forward(net, x):
    # networks/backbones/resnet.py:123-131
    x = F.relu(_bn(_cv(x, net.conv1), net.bn1))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for stage in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in stage:
            x = bottleneck_forward(blk, x)
    return x


def ppm_forward(dec, feats):
    # networks/pspnet_pop.py:31-35 (+ stage definition :25-29)
    h, w = feats.shape[2:]
    priors = []
    for size, st in zip(dec.sizes, dec.stages):
        p = F.adaptive_avg_pool2d(feats, (size, size))
        p = F.relu(_bn(_cv(p, st[1]), st[2]))
        priors.append(F.interpolate(p, size=(h, w), mode='bilinear', align_corners=False))
    cat = torch.cat(priors + [feats], 1)
    b = dec.bottleneck
    return _cv(F.relu(_bn(_cv(cat, b[0]), b[1])), b[3])


def orthogonal_decompose(feats, bases_b, bases_n=None):
    # networks/pspnet_pop.py:95-121; feats [B,C,N], bases [1,K,C]
    q = feats.to(torch.float)
    s1 = F.normalize(bases_b.to(torch.float), p=2, dim=-1)
    proj1 = torch.matmul(s1, q)                              # [B,K,N]
    out_fg_b = proj1.unsqueeze(2) * s1.unsqueeze(-1)         # [B,K,C,N]
    out_bg = q - out_fg_b.sum(1)
    if bases_n is None:
        return out_fg_b, out_bg.unsqueeze(1)
    s2 = F.normalize(bases_n, p=2, dim=-1)
    proj2 = torch.matmul(s2, q)
    out_fg_n = proj2.unsqueeze(2) * s2.unsqueeze(-1)
    out_bg = out_bg - out_fg_n.sum(1)
    return out_fg_b, out_fg_n, out_bg.unsqueeze(1)


def classifier_forward(cls, x):
    # networks/pspnet_pop.py:46-52: 1x1 -> ReLU -> 1x1 -> ReLU -> 1x1, no biases
    return _cv(F.relu(_cv(F.relu(_cv(x, cls[0])), cls[2])), cls[4])


def features_of(model, img):
    return ppm_forward(model.decoder, resnet_forward(model.backbone, img))


def head_base(model, features):
    # networks/pspnet_pop.py:171-182
    B, C, h, w = features.shape
    cls_emb = model.base_emb.unsqueeze(0)
    n_class = 1 + cls_emb.shape[1]
    fg, bg = orthogonal_decompose(features.flatten(2), cls_emb)
    allf = torch.cat([bg, fg], dim=1).contiguous().view(B * n_class, C, h, w)
    return classifier_forward(model.classifier, allf).view(B, n_class, h, w)


def head_all(model, features):
    # networks/pspnet_pop.py:143-160 / :201-219 -- channel order [bg | base | novel]
    B, C, h, w = features.shape
    fg_b, fg_n, bg = orthogonal_decompose(features.flatten(2), model.base_emb.unsqueeze(0), model.novel_emb.unsqueeze(0))
    preds1 = classifier_forward(model.classifier, fg_b.reshape(B * model.n_base, C, h, w)).view(B, model.n_base, h, w)
    feats_n = torch.cat([bg, fg_n], dim=1).reshape(B * (1 + model.n_novel), C, h, w)
    preds2 = classifier_forward(model.classifier_n, feats_n).view(B, 1 + model.n_novel, h, w)
    return torch.cat([preds2[:, 0].unsqueeze(1), preds1, preds2[:, 1:]], dim=1), preds2


def forward_base(model, img, mask=None):
    # networks/pspnet_pop.py:162-189
    preds = head_base(model, features_of(model, img))
    if model.criterion is not None and mask is not None:
        e = F.normalize(model.base_emb.unsqueeze(0), p=2, dim=-1).squeeze(0)
        return model.criterion(preds, mask, proto_sim=torch.matmul(e, e.t()))
    return preds


def forward_all(model, img):
    # networks/pspnet_pop.py:136-160
    return head_all(model, features_of(model, img))[0]


def pseudo_label(preds2_b, mask_b, n_base):
    """networks/pspnet_pop.py:221-231 for one base tile: upsample the [1+Kn,h,w] novel-head logits
    (align_corners=True), argmax, shift novel ids by n_base, overwrite mask_b where it is 0 (in place)."""
    bg_mask = mask_b == 0
    up = F.interpolate(preds2_b.unsqueeze(0), size=mask_b.shape, mode='bilinear', align_corners=True)
    idx = torch.argmax(up.squeeze(0), dim=0)
    idx[idx > 0] += n_base
    mask_b[bg_mask] = idx[bg_mask]
    return mask_b


def forward_novel(model, img, mask, img_b, mask_b):
    # networks/pspnet_pop.py:191-243
    full = torch.cat([img, img_b], dim=0)
    preds, preds2 = head_all(model, features_of(model, full))
    B = full.shape[0]
    new = [pseudo_label(preds2[B // 2 + b], mask_b[b], model.n_base) for b in range(B // 2)]
    mask_new = torch.stack(new, dim=0)
    if model.criterion is not None and mask is not None:
        C = model.novel_emb.shape[1]
        ne = F.normalize(model.novel_emb.unsqueeze(0).to(torch.float), p=2, dim=-1).reshape(-1, C)
        ae = torch.cat([ne, F.normalize(model.base_emb.to(torch.float), p=2, dim=-1)], dim=0)
        sim = torch.matmul(ne, ae.t())   # [Kn, Kn+Kb]
        return model.criterion(preds.to(torch.float), torch.cat([mask, mask_new], dim=0), is_ft=True, proto_sim=sim)
    return preds


# ----------------------------------------------------------------------------- loss
class OrthLossOracle(nn.Module):
    """loss/criterion.py:29-65."""

    def __init__(self, ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index
        self.w = 10.0

    def get_orth_loss(self, proto_sim):
        # criterion.py:37-43: mean |.| over the strict upper triangle (also of a rectangular [K1,K2])
        sel = torch.triu(torch.ones_like(proto_sim), diagonal=1) == 1
        return proto_sim[sel].abs().mean()

    def forward(self, preds, target, is_ft=False, proto_sim=None, aux_preds=None):
        up = F.interpolate(preds, size=target.shape[1:], mode='bilinear', align_corners=True)   # criterion.py:51
        seg = F.cross_entropy(up, target, ignore_index=self.ignore_index, reduction='mean')      # criterion.py:52
        orth = self.get_orth_loss(proto_sim)
        if aux_preds is not None:                                                                # criterion.py:56-60 (an auxiliary head's logits, weight 0.4)
            up_aux = F.interpolate(aux_preds, size=target.shape[1:], mode='bilinear', align_corners=True)
            aux = F.cross_entropy(up_aux, target, ignore_index=self.ignore_index, reduction='mean')
            return {'total_loss': seg + orth * self.w + 0.4 * aux, 'seg_loss': seg, 'aux_loss': aux, 'orth_loss': orth}
        return {'total_loss': seg + orth * self.w, 'seg_loss': seg, 'orth_loss': orth}           # criterion.py:61-63


# ----------------------------------------------------------------------------- dead-code row a-11
def masked_average_pooling(feature, mask):
    # networks/pspnet.py:7-15
    m = F.interpolate(mask, size=feature.shape[-2:], mode='bilinear', align_corners=True)
    pooled = torch.sum(feature * m, dim=(2, 3)) / (m.sum(dim=(2, 3)) + 1e-5)
    return pooled.mean(0, keepdim=True).unsqueeze(1)


# ----------------------------------------------------------------------------- driver helpers (a-12)
def intersection_and_union(output, target, K, ignore_index=255):
    """utils/pyt_utils.py:293-305 with bincount in place of histc (histc rejects int64 on CPU; counts are equal).
    Mutates ``output`` in place like the reference (ignored pixels are overwritten with ignore_index)."""
    output = output.reshape(-1)
    target = target.reshape(-1)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    def hist(v):
        v = v[(v >= 0) & (v <= K - 1)]
        return torch.bincount(v, minlength=K).to(torch.float32)
    a_i, a_o, a_t = hist(inter), hist(output), hist(target)
    return a_i, a_o + a_t - a_i, a_t


def confusion_matrix(gt_label, pred_label, class_num):
    """utils/pyt_utils.py:182-200 get_confusion_matrix: bincount of gt*K + pred over the pixels the caller kept (numpy int arrays)."""
    index = (np.asarray(gt_label).astype(np.int64) * class_num + np.asarray(pred_label).astype(np.int64)).astype('int32')
    count = np.bincount(index, minlength=class_num * class_num)[:class_num * class_num]
    return count.reshape(class_num, class_num).astype(np.float64)


def eval_confusion(logits, label, num_classes, ignore_label, pad_to_longside=False):
    """eval_base.py:166-177 (eval_ft.py:166-181 when pad_to_longside): upsample(align_corners=True) -> argmax -> drop ignore -> confusion."""
    h, w = label.shape[-2:]
    if pad_to_longside:
        side = max(h, w)
        out = F.interpolate(logits, size=(side, side), mode='bilinear', align_corners=True)
        gt = np.ones((label.shape[0], side, side), dtype=np.int64) * ignore_label
        gt[:, :h, :w] = label.numpy()
    else:
        out = F.interpolate(logits, size=(h, w), mode='bilinear', align_corners=True)
        gt = label.numpy().astype(np.int64)
    pred = np.asarray(np.argmax(out.numpy(), axis=1), dtype=np.uint8)
    keep = gt != ignore_label
    return pred, confusion_matrix(gt[keep], pred[keep], num_classes)


def miou_from_confusion(cm, n_base):
    """eval_base.py:193-199."""
    pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = tp / (pos + res - tp)
    return iou, np.nanmean(iou[:n_base + 1]), np.nanmean(iou[n_base + 1:]), np.nanmean(iou)


def param_groups(model, lr, scale=10.0):
    """utils/pyt_utils.py:216-249 (freeze_backbone=False branch): [backbone lr] [non-backbone bias lr*10 wd 0] [rest lr*10]."""
    wd0, lr1, lr10, keys = [], [], [], ([], [], [])
    for k, v in model.named_parameters():
        if not v.requires_grad:
            continue
        if 'backbone' not in k:
            if 'bias' in k:
                wd0.append(v); keys[1].append(k)
            else:
                lr10.append(v); keys[2].append(k)
        else:
            lr1.append(v); keys[0].append(k)
    groups = [{'params': lr1, 'lr': lr}, {'params': wd0, 'lr': lr * scale, 'weight_decay': 0.0},
              {'params': lr10, 'lr': lr * scale}]
    return groups, keys


def lr_poly(base_lr, it, max_it, power):
    # train_base.py:113-114
    return base_lr * ((1 - float(it) / max_it) ** power)


def train_step(model, optimizer, img, mask, clip_grad=5.0, double_step=True):
    """Loop body of train_base.py:250-264 on CPU fp32 (GradScaler disabled == identity):
    zero_grad, forward, backward, clip_grad_norm_(5.0), optimizer.step() -- and the reference's
    second optimizer.step() (train_base.py:264) when ``double_step``."""
    optimizer.zero_grad()
    loss = model(img, mask)
    loss['total_loss'].backward()
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
    optimizer.step()
    if double_step:
        optimizer.step()
    return {k: float(v) for k, v in loss.items()}, float(norm)


# ----------------------------------------------------------------------------- explicit index rules
# Closed-form restatements of the ATen index rules the HIP kernels implement; checked against
# F.adaptive_avg_pool2d / F.interpolate in tests/test_oracle_golden.py.
def adaptive_bins(in_size, out_size):
    """ATen adaptive_avg_pool start/end: [floor(i*in/out), ceil((i+1)*in/out))."""
    return [((i * in_size) // out_size, -((-(i + 1) * in_size) // out_size)) for i in range(out_size)]


def bilinear_taps(in_size, out_size, align_corners):
    """ATen upsample_bilinear2d source index rule (float32 arithmetic like the CPU kernel):
    returns (i0, i1, lambda1) per output index; value = (1-l1)*v[i0] + l1*v[i1]."""
    import numpy as np
    out = []
    if align_corners:
        scale = np.float32(in_size - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0)
    else:
        scale = np.float32(in_size) / np.float32(out_size)
    for d in range(out_size):
        if align_corners:
            src = scale * np.float32(d)
        else:
            src = scale * (np.float32(d) + np.float32(0.5)) - np.float32(0.5)
            if src < 0:
                src = np.float32(0)
        i0 = int(src)
        i1 = i0 + (1 if i0 < in_size - 1 else 0)
        out.append((i0, i1, float(np.float32(src) - np.float32(i0))))
    return out
