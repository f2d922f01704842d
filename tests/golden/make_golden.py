#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself (build container only).

Imports /root/reference (read-only, never copied) with stub modules for the packages the
image lacks, loads oracle/formula.py weights into both the reference model
and oracle/pop_oracle.py, asserts the two agree, and stores the reference's outputs.
Only data (inputs are regenerated from formulas; expected outputs are stored) lands here.

    python tests/golden/make_golden.py [g1 g2 ...]
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


class DropPathStandIn(nn.Module):
    """timm.models.layers.DropPath (timm is not installed here): x * mask / keep_prob with a per-sample Bernoulli(keep_prob) mask in train mode.
    The per-sample scale vector comes from `scale_fn(p, B)` so that the reference, the oracle and the HIP model can be fed the same draws; every
    instance numbers its calls in construction order (= the block order of the Swin backbone, two calls per block)."""
    scale_fn = None
    calls = 0

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0 or DropPathStandIn.scale_fn is None:
            return x
        s = DropPathStandIn.scale_fn(self.drop_prob, x.shape[0], self)
        return x if s is None else x * s.view(-1, *([1] * (x.dim() - 1)))


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub('cv2')
    stub('timm'); stub('timm.models')
    stub('timm.models.layers', DropPath=DropPathStandIn, to_2tuple=lambda x: (x, x),
         trunc_normal_=nn.init.trunc_normal_)
    stub('timm.models.registry', register_model=lambda f: f)
    stub('torchvision'); stub('torchvision.models')
    sys.path.insert(0, REF)
    import networks.pspnet_pop as ref_pop          # noqa
    import networks.pspnet as ref_psp              # noqa
    from loss.criterion import OrthLoss            # noqa
    import utils.pyt_utils as ref_utils            # noqa
    from networks.backbones.resnet import Bottleneck  # noqa
    return ref_pop, ref_psp, OrthLoss, ref_utils, Bottleneck


ref_pop, ref_psp, RefOrthLoss, ref_utils, RefBottleneck = import_reference()
from oracle import formula as fm            # noqa: E402
from oracle import pop_oracle as po         # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


def same(a, b, what, tol=0.0):
    a, b = a.detach(), b.detach()
    if tol == 0.0:
        assert torch.equal(a, b), '%s: oracle != reference (max abs %g)' % (what, (a - b).abs().max())
    else:
        assert torch.allclose(a, b, rtol=tol, atol=tol), '%s: max abs %g' % (what, (a - b).abs().max())


def build_pair(is_ft=False, n_novel=0, backbone='resnet50'):
    crit_r, crit_o = RefOrthLoss(ignore_index=255), po.OrthLossOracle(ignore_index=255)
    ref = ref_pop.GFSS_Model(n_base=7, criterion=crit_r, is_ft=is_ft, n_novel=n_novel, backbone=backbone,
                             pretrained_model=None, dilated=True, os=8)
    ora = po.PopOracle(n_base=7, criterion=crit_o, is_ft=is_ft, n_novel=n_novel, backbone=backbone)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), 'state_dict keys differ'
    assert [k for k, _ in ref.named_parameters()] == [k for k, _ in ora.named_parameters()]
    sd = fm.formula_state_dict(ref)
    ref.load_state_dict(sd, strict=True)
    ora.load_state_dict(sd, strict=True)
    return ref, ora


# ------------------------------------------------------------------------------------------ G1
def g1():
    feats = fm.sym('g1/feats', (2, 512, 24), 1.0)
    bb, bn = fm.sym('g1/bb', (1, 7, 512), 1.0), fm.sym('g1/bn', (1, 4, 512), 1.0)
    ref, _ = build_pair()
    fg, bg = ref.orthogonal_decompose(feats, bb)
    fg_o, bg_o = po.orthogonal_decompose(feats, bb)
    same(fg, fg_o, 'g1 fg'); same(bg, bg_o, 'g1 bg')
    fgb2, fgn2, bg2 = ref.orthogonal_decompose(feats, bb, bn)
    o = po.orthogonal_decompose(feats, bb, bn)
    same(fgb2, o[0], 'g1 fgb2'); same(fgn2, o[1], 'g1 fgn2'); same(bg2, o[2], 'g1 bg2')
    s1 = F.normalize(bb, p=2, dim=-1)
    save('g1_decompose', proj=torch.matmul(s1, feats), fg_sub=fg[:, :, ::32], bg=bg,
         fgn_sub=fgn2[:, :, ::32], bg2=bg2)


# ------------------------------------------------------------------------------------------ G2
def _head_only(model):
    model.backbone.base_forward = lambda x: x
    model.decoder = nn.Identity()
    return model


def g2():
    ref, ora = build_pair()
    _head_only(ref)
    feats = fm.sym('g2/feats', (2, 512, 8, 8), 1.0).requires_grad_(True)
    coef = fm.sym('g2/coef', (2, 8, 8, 8), 1.0)
    ref.criterion = None
    preds = ref(feats)
    (preds * coef).sum().backward()
    feats_o = feats.detach().clone().requires_grad_(True)
    preds_o = po.head_base(ora, feats_o)
    (preds_o * coef).sum().backward()
    same(preds, preds_o, 'g2 preds')
    same(feats.grad, feats_o.grad, 'g2 dfeats', 1e-6)
    same(ref.base_emb.grad, ora.base_emb.grad, 'g2 demb', 1e-6)
    save('g2_head', preds=preds, dfeats=feats.grad, d_base_emb=ref.base_emb.grad,
         d_cls0=ref.classifier[0].weight.grad[::8, ::8, 0, 0], d_cls2=ref.classifier[2].weight.grad[::8, ::8, 0, 0],
         d_cls4=ref.classifier[4].weight.grad[0, :, 0, 0])
    # ft head (forward_all) on the same feats
    ref, ora = build_pair(is_ft=True, n_novel=4)
    _head_only(ref)
    ref.eval(); ora.eval()
    pa = ref(feats.detach())
    pa_o, _ = po.head_all(ora, feats.detach())
    same(pa, pa_o, 'g2 preds_all')
    save('g2_head_all', preds=pa)


# ------------------------------------------------------------------------------------------ G3
def g3():
    cr, co = RefOrthLoss(ignore_index=255), po.OrthLossOracle(ignore_index=255)
    preds = fm.sym('g3/preds', (2, 8, 8, 8), 2.0).requires_grad_(True)
    target = fm.formula_mask(2, 64, 64, 8, tag='g3/mask', block=8, ignore_rows=5)
    e = F.normalize(fm.sym('g3/emb', (7, 512), 1.0), dim=-1)
    sim = (e @ e.t()).requires_grad_(True)
    d = cr(preds, target, proto_sim=sim)
    d['total_loss'].backward()
    po_preds = preds.detach().clone().requires_grad_(True)
    d_o = co(po_preds, target, proto_sim=sim.detach())
    d_o['total_loss'].backward()
    for k in d:
        same(d[k], d_o[k], 'g3 ' + k)
    same(preds.grad, po_preds.grad, 'g3 dpreds')
    rect = fm.sym('g3/rect', (4, 11), 1.0).requires_grad_(True)
    orth_rect = cr.get_orth_loss(rect, is_ft=True)
    orth_rect.backward()
    same(orth_rect, co.get_orth_loss(rect.detach()), 'g3 rect')
    # 12-class ft-shaped case with labels 0..11 and 255
    preds12 = fm.sym('g3/preds12', (2, 12, 8, 8), 2.0).requires_grad_(True)
    t12 = fm.formula_mask(2, 64, 64, 12, tag='g3/mask12', block=8, ignore_rows=3)
    d12 = cr(preds12, t12, is_ft=True, proto_sim=rect.detach())
    d12['total_loss'].backward()
    save('g3_loss', total=d['total_loss'], seg=d['seg_loss'], orth=d['orth_loss'], dpreds=preds.grad, dsim=sim.grad,
         orth_rect=orth_rect, d_rect=rect.grad, total12=d12['total_loss'], seg12=d12['seg_loss'], dpreds12=preds12.grad)


# ------------------------------------------------------------------------------------------ G3b
def g3b():
    """criterion.py:56-60: the aux_preds branch of OrthLoss.forward (total = seg + 10 orth + 0.4 aux; four entries in the dict)."""
    cr, co = RefOrthLoss(ignore_index=255), po.OrthLossOracle(ignore_index=255)
    preds = fm.sym('g3b/preds', (2, 8, 8, 8), 2.0).requires_grad_(True)
    aux = fm.sym('g3b/aux', (2, 8, 16, 16), 1.5).requires_grad_(True)
    target = fm.formula_mask(2, 64, 64, 8, tag='g3b/mask', block=8, ignore_rows=7)
    e = F.normalize(fm.sym('g3b/emb', (7, 512), 1.0), dim=-1)
    sim = (e @ e.t()).requires_grad_(True)
    d = cr(preds, target, proto_sim=sim, aux_preds=aux)
    d['total_loss'].backward()
    p2, a2, s2 = (t.detach().clone().requires_grad_(True) for t in (preds, aux, sim))
    d_o = co(p2, target, proto_sim=s2, aux_preds=a2)
    d_o['total_loss'].backward()
    assert sorted(d) == sorted(d_o) == ['aux_loss', 'orth_loss', 'seg_loss', 'total_loss']
    for k in d:
        same(d[k], d_o[k], 'g3b ' + k)
    same(preds.grad, p2.grad, 'g3b dpreds'); same(aux.grad, a2.grad, 'g3b daux'); same(sim.grad, s2.grad, 'g3b dsim')
    save('g3b_loss_aux', total=d['total_loss'], seg=d['seg_loss'], aux=d['aux_loss'], orth=d['orth_loss'], dpreds=preds.grad, daux=aux.grad, dsim=sim.grad)


# ------------------------------------------------------------------------------------------ G4
def g4():
    for tag, (feat, outf, hw) in {'a': (64, 64, 12), 'b': (128, 64, 16)}.items():
        ref = ref_pop.PSPModule(feat, out_features=outf)
        ora = po.make_ppm(feat, outf)
        sd = {k: fm.formula_tensor('g4' + tag + '/' + k, v) for k, v in ref.state_dict().items()}
        ref.load_state_dict(sd); ora.load_state_dict(sd)
        x = fm.sym('g4%s/x' % tag, (2, feat, hw, hw), 1.0).relu_().requires_grad_(True)
        coef = fm.sym('g4%s/coef' % tag, (2, outf, hw, hw), 1.0)
        ref.train(); ora.train()
        y = ref(x); (y * coef).sum().backward()
        xo = x.detach().clone().requires_grad_(True)
        yo = po.ppm_forward(ora, xo); (yo * coef).sum().backward()
        same(y, yo, 'g4 y'); same(x.grad, xo.grad, 'g4 dx', 1e-6)
        ref.eval()
        ye = ref(x.detach())
        save('g4_ppm_' + tag, y=y, dx=x.grad, y_eval=ye,
             d_bott_w=ref.bottleneck[0].weight.grad[::4, ::16], d_stage3_w=ref.stages[3][1].weight.grad[:, :, 0, 0],
             d_stage0_gamma=ref.stages[0][2].weight.grad, rm_stage3=ref.stages[3][2].running_mean,
             rv_bott=ref.bottleneck[1].running_var)


# ------------------------------------------------------------------------------------------ G5
G5_CASES = {  # name: (inplanes, planes, stride, dilation, downsample)
    's1_ds': (64, 64, 1, 1, True),
    's1_id': (256, 64, 1, 1, False),
    's2_ds': (256, 128, 2, 1, True),
    'd2_ds': (512, 256, 1, 2, True),
    'd4_id': (1024, 256, 1, 4, False),
}


def g5():
    for name, (inp, pl, st, dil, ds) in G5_CASES.items():
        dsm = None
        if ds:
            dsm = nn.Sequential(nn.Conv2d(inp, pl * 4, 1, stride=st, bias=False), nn.BatchNorm2d(pl * 4))
        ref = RefBottleneck(inp, pl, stride=st, dilation=dil, downsample=dsm)
        ora = po.make_bottleneck(inp, pl, st, dil, ds)
        sd = {k: fm.formula_tensor('g5' + name + '/' + k, v) for k, v in ref.state_dict().items()}
        ref.load_state_dict(sd); ora.load_state_dict(sd)
        x = fm.sym('g5%s/x' % name, (2, inp, 16, 16), 1.0).relu_().requires_grad_(True)
        ref.train(); ora.train()
        y = ref(x)
        coef = fm.sym('g5%s/coef' % name, tuple(y.shape), 1.0)
        (y * coef).sum().backward()
        xo = x.detach().clone().requires_grad_(True)
        yo = po.bottleneck_forward(ora, xo); (yo * coef).sum().backward()
        same(y, yo, 'g5 y ' + name); same(x.grad, xo.grad, 'g5 dx ' + name, 1e-6)
        same(ref.bn2.running_var, ora.bn2.running_var, 'g5 rv')
        ref.eval()
        ye = ref(x.detach())
        save('g5_bottleneck_' + name, y=y[:, ::4], dx=x.grad[:, ::4], y_eval=ye[:, ::4],
             d_conv2_w=ref.conv2.weight.grad[::4, ::4], d_conv1_w=ref.conv1.weight.grad[::4, ::4, 0, 0],
             d_conv3_w=ref.conv3.weight.grad[::8, ::4, 0, 0],
             d_bn3_gamma=ref.bn3.weight.grad, d_bn1_beta=ref.bn1.bias.grad,
             rm_bn2=ref.bn2.running_mean, rv_bn2=ref.bn2.running_var)


# ------------------------------------------------------------------------------------------ G6
def g6():
    ref, ora = build_pair()
    img = fm.formula_image(2, 512, 512, 'g6/img')
    mask = fm.formula_mask(2, 512, 512, 8, 'g6/mask')
    ref.train(); ora.train()
    crit = ref.criterion
    ref.criterion = None
    logits = ref(img)                          # train-mode forward (batch-stat BN), updates running stats
    ref.criterion = crit
    sim_e = F.normalize(ref.base_emb.unsqueeze(0), p=2, dim=-1).squeeze(0)
    loss = crit(logits, mask, proto_sim=sim_e @ sim_e.t())
    loss['total_loss'].backward()
    gnorm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1e30)
    lo = ora(img, mask)
    lo['total_loss'].backward()
    gnorm_o = torch.nn.utils.clip_grad_norm_(ora.parameters(), 1e30)
    for k in loss:
        same(loss[k], lo[k], 'g6 ' + k)
    same(gnorm, gnorm_o, 'g6 gnorm', 1e-6)
    same(ref.backbone.conv1.weight.grad, ora.backbone.conv1.weight.grad, 'g6 dconv1', 1e-6)
    up = F.interpolate(logits, size=(512, 512), mode='bilinear', align_corners=True)
    amax = up.argmax(1).to(torch.uint8)
    ref.eval()
    logits_eval = ref(img)                     # eval-mode forward with the just-updated running stats
    gn = {k: p.grad.norm() for k, p in ref.named_parameters()}
    keys = sorted(gn)
    save('g6_full_r50', logits=logits, argmax=amax, total=loss['total_loss'], seg=loss['seg_loss'], orth=loss['orth_loss'],
         gnorm=gnorm, d_base_emb=ref.base_emb.grad, d_cls4=ref.classifier[4].weight.grad[0, :, 0, 0],
         d_conv1=ref.backbone.conv1.weight.grad, d_dec_bias=ref.decoder.bottleneck[3].bias.grad,
         rm_bn1=ref.backbone.bn1.running_mean, rv_bn1=ref.backbone.bn1.running_var,
         rm_l4=ref.backbone.layer4[2].bn3.running_mean,
         logits_eval=logits_eval, grad_norm_keys=np.array(keys), grad_norms=torch.stack([gn[k] for k in keys]))


# ------------------------------------------------------------------------------------------ G7
def g7():
    ref, ora = build_pair(is_ft=True, n_novel=4)
    ref.init_cls_n(); po.init_cls_n(ora)
    # make classifier_n differ from classifier so channel routing mistakes show
    with torch.no_grad():
        for (k, p), (_, q) in zip(ref.classifier_n.named_parameters(), ora.classifier_n.named_parameters()):
            p.add_(fm.sym('g7/cn/' + k, tuple(p.shape), 0.01)); q.copy_(p)
    img = fm.formula_image(1, 512, 512, 'g7/img')
    img_b = fm.formula_image(1, 512, 512, 'g7/img_b')
    mask = fm.formula_mask(1, 512, 512, 4, 'g7/mask', ignore_rows=0, lo=8)     # novel ids 8..11
    mask[mask == 8] = 255                                                     # some ignore (oem_ft.py:197 style)
    mask_b = fm.formula_mask(1, 512, 512, 8, 'g7/mask_b', ignore_rows=0)       # base ids 0..7
    mb_r, mb_o = mask_b.clone(), mask_b.clone()
    ref.train_mode(); po.train_mode(ora)
    d = ref(img, mask, img_b, mb_r)
    d['total_loss'].backward()
    do = ora(img, mask, img_b, mb_o)
    do['total_loss'].backward()
    for k in d:
        same(d[k], do[k], 'g7 ' + k)
    assert torch.equal(mb_r, mb_o), 'g7 pseudo labels differ'
    same(ref.novel_emb.grad, ora.novel_emb.grad, 'g7 dnovel', 1e-6)
    crit = ref.criterion
    ref.criterion = None
    preds = ref(img, mask, img_b, mask_b.clone())
    ref.criterion = crit
    ref.eval()
    pall = ref(img)
    save('g7_ft', preds=preds, mask_b_new=mb_r.to(torch.uint8), total=d['total_loss'], seg=d['seg_loss'], orth=d['orth_loss'],
         d_novel_emb=ref.novel_emb.grad, d_clsn4=ref.classifier_n[4].weight.grad[0, :, 0, 0],
         d_clsn0=ref.classifier_n[0].weight.grad[::8, ::8, 0, 0], preds_all=pall)


# ------------------------------------------------------------------------------------------ G8
def g8():
    ref, ora = build_pair()
    img = fm.formula_image(2, 512, 512, 'g6/img')
    mask = fm.formula_mask(2, 512, 512, 8, 'g6/mask')
    groups = ref_utils.get_parameters(ref, lr=1e-5)
    groups_o, keys_o = po.param_groups(ora, lr=1e-5)
    sizes = [(len(g['params']), sum(p.numel() for p in g['params'])) for g in groups]
    sizes_o = [(len(g['params']), sum(p.numel() for p in g['params'])) for g in groups_o]
    assert sizes == sizes_o, (sizes, sizes_o)
    opt = torch.optim.AdamW(groups, lr=1e-5, weight_decay=1e-4)
    ref.train()
    losses, norms = [], []
    for step in range(3):
        # loop body of train_base.py:250-264 with GradScaler as identity (CPU): the scaler's step + the
        # explicit optimizer.step() == two AdamW steps on the same grads
        opt.zero_grad()
        d = ref(img, mask)
        d['total_loss'].backward()
        norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        opt.step(); opt.step()
        losses.append([float(d['total_loss']), float(d['seg_loss']), float(d['orth_loss'])]); norms.append(float(norm))
        print('g8 step', step, losses[-1], norms[-1])
    save('g8_traj', losses=np.array(losses, dtype=np.float64), norms=np.array(norms, dtype=np.float64),
         group_sizes=np.array(sizes, dtype=np.int64), base_emb_after=ref.base_emb.detach(),
         group_keys0=np.array(keys_o[0]), group_keys1=np.array(keys_o[1]), group_keys2=np.array(keys_o[2]))


# ------------------------------------------------------------------------------------------ G9 / G10
def g9():
    feat = fm.sym('g9/feat', (2, 32, 8, 8), 1.0)
    m = (fm.uniform01('g9/mask', 2 * 64 * 64).reshape(2, 1, 64, 64) > 0.5).float()
    r = ref_psp.masked_average_pooling(feat, m)
    same(r, po.masked_average_pooling(feat, m), 'g9')
    save('g9_map', proto=r)


def g10():
    pred = (fm.uniform01('g10/pred', 2 * 64 * 64) * 8).floor().long().reshape(2, 64, 64)
    tgt = (fm.uniform01('g10/tgt', 2 * 64 * 64) * 8).floor().long().reshape(2, 64, 64)
    tgt[0, :5] = 255
    # the reference's histc rejects int64 on CPU; run it on float copies -- same counts
    i, u, t = ref_utils.intersectionAndUnionGPU(pred.clone().float(), tgt.clone().float(), 8, 255)
    io, uo, to = po.intersection_and_union(pred.clone(), tgt.clone(), 8, 255)
    same(i, io, 'g10 i'); same(u, uo, 'g10 u'); same(t, to, 'g10 t')
    save('g10_iou', inter=i, union=u, target=t)


def g11():
    """eval path: eval_base.py:166-199 on fixed logits / labels, both the plain and the eval_ft long-side variant."""
    logits = fm.sym('g11/logits', (2, 12, 16, 12), 2.0)
    label = (fm.uniform01('g11/label', 2 * 128 * 96) * 12).floor().long().reshape(2, 128, 96)
    label[1, :9] = 255
    for tag, pad in (('plain', False), ('ft', True)):
        h, w = label.shape[-2:]
        side = max(h, w)
        size = (side, side) if pad else (h, w)
        out = F.interpolate(logits, size=size, mode='bilinear', align_corners=True)
        seg_pred = np.asarray(np.argmax(out.numpy(), axis=1), dtype=np.uint8)
        seg_gt = label.numpy().astype(np.int64)
        if pad:
            pad_gt = np.ones((2, side, side), dtype=np.int64) * 255
            pad_gt[:, :h, :w] = seg_gt
            seg_gt = pad_gt
        keep = seg_gt != 255
        cm = ref_utils.get_confusion_matrix(seg_gt[keep], seg_pred[keep], 12)
        pred_o, cm_o = po.eval_confusion(logits, label, 12, 255, pad_to_longside=pad)
        assert np.array_equal(pred_o, seg_pred) and np.array_equal(cm, cm_o), 'g11 ' + tag
        pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
        iou = tp / (pos + res - tp)
        same(torch.tensor(iou), torch.tensor(po.miou_from_confusion(cm, 7)[0]), 'g11 iou')
        save('g11_eval_' + tag, pred=seg_pred, cm=cm, iou=iou, miou=np.array([np.nanmean(iou[:8]), np.nanmean(iou[8:]), np.nanmean(iou)]))


def g12():
    """Constructor kwargs off the default path (networks/backbones/resnet.py:81-121): multi_grid=(1,2,4) dilations in layer4, relu_l3 /
    relu_l4 = False (no final ReLU in the last block of the layer), at os 8 and os 16; full model, small tile."""
    for tag, kw in (('a', dict(dilated=True, os=8, multi_grid=True, relu_l3=True, relu_l4=False)),
                    ('b', dict(dilated=True, os=16, multi_grid=True, relu_l3=False, relu_l4=False))):
        ref = ref_pop.GFSS_Model(n_base=7, criterion=RefOrthLoss(ignore_index=255), backbone='resnet50', pretrained_model=None, **kw)
        ora = po.PopOracle(n_base=7, criterion=po.OrthLossOracle(ignore_index=255), backbone='resnet50', **kw)
        assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
        sd = fm.formula_state_dict(ref)
        ref.load_state_dict(sd, strict=True); ora.load_state_dict(sd, strict=True)
        img = fm.formula_image(2, 96, 128, 'g12%s/img' % tag)
        mask = fm.formula_mask(2, 96, 128, 8, 'g12%s/mask' % tag, ignore_rows=5)
        ref.train(); ora.train()
        lr, lo = ref(img, mask), ora(img, mask)
        lr['total_loss'].backward(); lo['total_loss'].backward()
        for k in lr:
            same(lr[k], lo[k], 'g12 ' + k)
        same(ref.base_emb.grad, ora.base_emb.grad, 'g12 d_base_emb', 1e-6)
        ref.eval(); ora.eval()
        with torch.no_grad():
            pe, po_ = ref(img), ora(img)
        same(pe, po_, 'g12 eval logits', 1e-6)
        save('g12_kwargs_' + tag, total=lr['total_loss'], seg=lr['seg_loss'], orth=lr['orth_loss'], d_base_emb=ref.base_emb.grad,
             logits_eval=pe, rm_l4=ref.backbone.layer4[2].bn3.running_mean)


# ------------------------------------------------------------------------------------------ G13-G16: Swin-POP
import networks.swin_pop as ref_swin                      # noqa: E402
import networks.backbones.swintransformer as ref_st       # noqa: E402
from oracle import swin_oracle as so                       # noqa: E402


def drop_scale(index, b, p):
    """Deterministic DropPath draw shared by the reference stand-in, the oracle and the HIP tests: sample b of block `index` is dropped when
    (7 * index + 3 * b) % 5 == 0, else scaled by 1 / (1 - p)."""
    return 0.0 if (7 * index + 3 * b) % 5 == 0 else 1.0 / (1.0 - p)


def _formula_load(mod, prefix):
    mod.load_state_dict({k: fm.formula_tensor(prefix + k, v) for k, v in mod.state_dict().items()}, strict=True)
    return mod


def g13():
    """One Swin stage = BasicLayer (swintransformer.py:293-392): a W-MSA block, a shifted SW-MSA block (mask), PatchMerging; token maps that
    need window padding (10 x 13 -> 14 x 14: pad tokens carry the qkv bias) and odd-size merge padding.  Case b: C = 192 (no channel pad), 7 x 7."""
    for tag, dim, heads, H, W in (('a', 96, 3, 10, 13), ('b', 192, 6, 7, 7)):
        ref = ref_st.BasicLayer(dim=dim, depth=2, num_heads=heads, window_size=7, drop_path=[0.0, 0.0], downsample=ref_st.PatchMerging)
        net = so._Box()
        st = so._Box(); st.blocks = nn.ModuleList()
        for j in range(2):
            blk = so.make_block(dim, heads); blk.shift, blk.drop_path_p, blk.index = (0 if j == 0 else 3), 0.0, j
            st.blocks.append(blk)
        st.downsample = so._Box(); st.downsample.reduction = nn.Linear(4 * dim, 2 * dim, bias=False); st.downsample.norm = nn.LayerNorm(4 * dim)
        assert list(ref.state_dict().keys()) == list(st.state_dict().keys())
        _formula_load(ref, 'g13%s/' % tag); _formula_load(st, 'g13%s/' % tag)
        x = fm.sym('g13%s/x' % tag, (2, H * W, dim), 1.0)
        c1 = fm.sym('g13%s/c1' % tag, (2, H * W, dim), 1.0)
        c2 = fm.sym('g13%s/c2' % tag, (2, ((H + 1) // 2) * ((W + 1) // 2), 2 * dim), 1.0)
        xr = x.clone().requires_grad_(True)
        xo_, _, _, xd, _, _ = ref(xr, H, W)
        ((xo_ * c1).sum() + (xd * c2).sum()).backward()
        holder = types.SimpleNamespace(drop_path_scale=lambda i, B, p: None)
        xo = x.clone().requires_grad_(True)
        Hp, Wp = -(-H // 7) * 7, -(-W // 7) * 7
        y = xo
        for blk in st.blocks:
            y = so.block_forward(holder, blk, y, H, W, so.shift_mask(Hp, Wp, 7, 3))
        yd = so.patch_merging(st.downsample, y, H, W)
        ((y * c1).sum() + (yd * c2).sum()).backward()
        same(xo_, y, 'g13 x_out'); same(xd, yd, 'g13 x_down'); same(xr.grad, xo.grad, 'g13 dx', 1e-6)
        pr, po_ = dict(ref.named_parameters()), dict(st.named_parameters())
        for k in pr:
            same(pr[k].grad, po_[k].grad, 'g13 d ' + k, 1e-5)
        sub = lambda t: t[::4, ::4] if (t.dim() == 2 and t.numel() > 20000) else t          # big weight gradients: every 4th row / column
        save('g13_swin_stage_' + tag, x_out=xo_[:, :, ::2], x_down=xd[:, :, ::2], dx=xr.grad[:, :, ::2],
             **{'d_' + k.replace('.', '_'): sub(pr[k].grad) for k in ('blocks.0.attn.qkv.bias', 'blocks.1.attn.qkv.bias', 'blocks.1.attn.relative_position_bias_table',
                                                                  'blocks.0.norm1.weight', 'blocks.1.norm2.bias', 'blocks.1.mlp.fc1.weight', 'blocks.0.attn.proj.weight',
                                                                  'blocks.1.attn.qkv.weight', 'blocks.0.mlp.fc2.bias', 'downsample.reduction.weight', 'downsample.norm.weight')})


def g14():
    """PatchEmbed (swintransformer.py:395-433) on an image whose sides are not multiples of 4 (zero padding) + LayerNorm."""
    ref = ref_st.PatchEmbed(patch_size=4, in_chans=3, embed_dim=96, norm_layer=nn.LayerNorm)
    _formula_load(ref, 'g14/')
    img = fm.formula_image(2, 30, 37, 'g14/img')
    coef = fm.sym('g14/coef', (2, 96, 8, 10), 1.0)
    y = ref(img)
    (y * coef).sum().backward()
    pe = so._Box(); pe.proj = nn.Conv2d(3, 96, 4, stride=4); pe.norm = nn.LayerNorm(96)
    _formula_load(pe, 'g14/')
    x = F.conv2d(F.pad(img, (0, 3, 0, 2)), pe.proj.weight, pe.proj.bias, stride=4)
    yo = so._ln(x.flatten(2).transpose(1, 2), pe.norm).transpose(1, 2).reshape(2, 96, 8, 10)
    (yo * coef).sum().backward()
    same(y, yo, 'g14 y'); same(ref.proj.weight.grad, pe.proj.weight.grad, 'g14 dw', 1e-5)
    save('g14_patch_embed', y=y, dw=ref.proj.weight.grad, db=ref.proj.bias.grad, dgamma=ref.norm.weight.grad, dbeta=ref.norm.bias.grad)


def g15():
    """UperNet_Decoder_Plus (swin_pop.py:104-173) with the Swin-T widths on small maps whose sizes do NOT double from level to level (top-down
    and final interpolations at non-2x ratios), train-mode BatchNorm, Dropout2d mask drawn by nn.Dropout2d under a fixed seed and stored."""
    filters = [96, 192, 384, 768]
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3)]
    ref = ref_swin.UperNet_Decoder_Plus(filters, 96)
    ora = so.SwinPopOracle(7)
    assert list(ref.state_dict().keys()) == list(ora.decoder.state_dict().keys())
    _formula_load(ref, 'g15/'); _formula_load(ora.decoder, 'g15/')
    xs = [fm.sym('g15/x%d' % i, (2, c, h, w), 1.0) for i, (c, (h, w)) in enumerate(zip(filters, sizes))]
    coef = fm.sym('g15/coef', (2, 96, 16, 20), 1.0)
    xr = [x.clone().requires_grad_(True) for x in xs]
    xo = [x.clone().requires_grad_(True) for x in xs]
    ref.train(); ora.train()
    masks = []
    default = ora._default_dropout2d
    ora.dropout2d_scale = lambda B, C, p: masks.append(default(B, C, p)) or masks[-1]
    torch.manual_seed(15); yr = ref(xr)
    torch.manual_seed(15); yo = so.decoder_forward(ora, xo)
    (yr * coef).sum().backward(); (yo * coef).sum().backward()
    same(yr, yo, 'g15 y')
    for a, b in zip(xr, xo):
        same(a.grad, b.grad, 'g15 dx', 1e-6)
    pr, po_ = dict(ref.named_parameters()), dict(ora.decoder.named_parameters())
    for k in pr:
        same(pr[k].grad, po_[k].grad, 'g15 d ' + k, 1e-5)
    ref.eval()
    with torch.no_grad():
        ye = ref([x for x in xs])
    save('g15_upernet', y=yr, y_eval=ye, drop_mask=masks[0], dx0=xr[0].grad[:, ::4], dx1=xr[1].grad[:, ::8], dx2=xr[2].grad[:, ::8], dx3=xr[3].grad[:, ::16],
         d_lat0_w=pr['lateral_convs.0.0.weight'].grad[::4, ::4], d_lat2_b=pr['lateral_convs.2.0.bias'].grad, d_fpn3_w=pr['fpn_convs.3.4.0.weight'].grad[::4, ::4],
         d_fpn0_gamma=pr['fpn_convs.0.0.1.weight'].grad, d_psp_bott_w=pr['psp.bottleneck.0.weight'].grad[::2, ::16, 0, 0], d_psp_st0_w=pr['psp.stages.0.1.weight'].grad[::4, ::16, 0, 0],
         d_psp_st3_gamma=pr['psp.stages.3.2.weight'].grad, rm_lat1=ref.lateral_convs[1][1].running_mean, rv_psp_bott=ref.psp.bottleneck[1].running_var,
         rm_fpn2=ref.fpn_convs[2][2][1].running_mean)


def g16():
    """Full Swin-T POP (BASELINE config 5's model) on a 128 x 160 tile pair: eval logits; one TRAIN-mode step with DropPath (shared deterministic
    draws through the timm stand-in) and Dropout2d (nn.Dropout2d's own draw, stored): loss dict, gradients, per-parameter gradient norms."""
    ref = ref_swin.GFSS_Model(n_base=7, criterion=RefOrthLoss(ignore_index=255), backbone='swin-t', pretrained_model=None)
    ora = so.SwinPopOracle(7, criterion=po.OrthLossOracle(255), backbone='swin-t')
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), 'swin state_dict keys differ'
    assert [k for k, _ in ref.named_parameters()] == [k for k, _ in ora.named_parameters()]
    sd = fm.formula_state_dict(ref)
    ref.load_state_dict(sd, strict=True); ora.load_state_dict(sd, strict=True)
    img = fm.formula_image(2, 128, 160, 'g16/img'); mask = fm.formula_mask(2, 128, 160, 8, 'g16/mask', ignore_rows=5)
    ref.eval(); ora.eval()
    with torch.no_grad():
        le, lo = ref(img), ora(img)
    same(le, lo, 'g16 eval logits')
    # train step.  The stand-in numbers DropPath calls through the module's drop_prob: p -> block index via the reference's linspace schedule
    rates = torch.linspace(0, 0.2, 12).tolist()
    index_of = lambda p: min(range(12), key=lambda i: abs(rates[i] - p))
    DropPathStandIn.scale_fn = lambda p, B, mod: torch.tensor([drop_scale(index_of(p), b, p) for b in range(B)])
    ora.drop_path_scale = lambda i, B, p: None if p <= 0.0 else torch.tensor([drop_scale(i, b, p) for b in range(B)])
    masks = []
    default = ora._default_dropout2d
    ora.dropout2d_scale = lambda B, C, p: masks.append(default(B, C, p)) or masks[-1]
    ref.train(); ora.train()
    torch.manual_seed(16); dr = ref(img, mask)
    torch.manual_seed(16); do = ora(img, mask)
    dr['total_loss'].backward(); do['total_loss'].backward()
    DropPathStandIn.scale_fn = None
    for k in dr:
        same(dr[k], do[k], 'g16 ' + k)
    pr, po_ = dict(ref.named_parameters()), dict(ora.named_parameters())
    for k in pr:
        same(pr[k].grad, po_[k].grad, 'g16 d ' + k, 1e-5)
    keys = [k for k in pr]
    save('g16_swin_pop', logits_eval=le, total=dr['total_loss'], seg=dr['seg_loss'], orth=dr['orth_loss'], drop_mask=masks[0],
         d_base_emb=pr['base_emb'].grad, d_cls4=pr['classifier.4.weight'].grad[0, :, 0, 0], d_patch_w=pr['backbone.patch_embed.proj.weight'].grad,
         d_table=pr['backbone.layers.0.blocks.1.attn.relative_position_bias_table'].grad, d_fc1=pr['backbone.layers.2.blocks.3.mlp.fc1.weight'].grad[::16, ::8],
         d_qkv_b=pr['backbone.layers.1.blocks.0.attn.qkv.bias'].grad, d_fpn3=pr['decoder.fpn_convs.3.4.0.weight'].grad[::4, ::4],
         grad_norm_keys=np.array(keys), grad_norms=np.array([float(pr[k].grad.norm()) for k in keys], dtype=np.float32))


# ------------------------------------------------------------------------------------------ G17: OEM tile preparation
def g17():
    """dataset/base_dataset.py crop / pad / random_flip / fixed_random_rotate / normalize / totensor in the order of oem.py:70-75 on synthetic
    tiles (smaller than, equal to and larger than the crop), with the reference's own random draws under fixed seeds; the label re-indexing of
    oem.py:113-133 through GFSSegVal.__getitem__ fed by a rasterio stand-in.  OpenCV is absent: cv2.copyMakeBorder (constant border) is its
    numpy equivalent here, nothing else of cv2 is on this path."""
    import importlib.util
    import random
    from oracle import data_oracle as do

    def copy_make_border(src, top, bottom, left, right, border_type, value=0):
        v = value[0] if isinstance(value, (tuple, list)) else value
        pads = ((top, bottom), (left, right)) + (((0, 0),) if src.ndim == 3 else ())
        return np.pad(src, pads, constant_values=v)
    sys.modules['cv2'].copyMakeBorder = copy_make_border
    sys.modules['cv2'].BORDER_CONSTANT = 0
    arrays = {}
    rio = types.ModuleType('rasterio')
    rio.open = lambda path: types.SimpleNamespace(read=lambda: arrays[path])
    sys.modules['rasterio'] = rio

    def load(name):
        spec = importlib.util.spec_from_file_location('ref_dataset.' + name, os.path.join(REF, 'dataset', name + '.py'), submodule_search_locations=None)
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = 'ref_dataset'
        sys.modules['ref_dataset.' + name] = mod
        spec.loader.exec_module(mod)
        return mod
    pkg = types.ModuleType('ref_dataset'); pkg.__path__ = [os.path.join(REF, 'dataset')]
    sys.modules['ref_dataset'] = pkg
    base = load('base_dataset')
    oem = load('oem')
    out = {}
    for tag, (H, W) in (('small', (50, 70)), ('exact', (64, 64)), ('large', (100, 90))):
        img = (fm.uniform01('g17/%s/img' % tag, H * W * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(H, W, 3).numpy()
        lab = (fm.uniform01('g17/%s/lab' % tag, H * W) * 12).floor().to(torch.uint8).reshape(H, W).numpy()
        lab[:7] = 255
        ds = base.BaseDataset(mode='train', crop_size=(64, 64), ignore_label=255, base_size=(1024, 1024))
        ds.mean, ds.std = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]                 # oem.py:26-27
        for rep in range(3):
            seed = 100 * rep + H
            random.seed(seed); np.random.seed(seed)
            i, l = ds.crop(img, lab)
            i, l = ds.pad(ds.crop_size, i, l)
            i, l = ds.random_flip(i, l)
            i, l = ds.fixed_random_rotate(i, l)
            i = ds.normalize(i)
            it, lt = ds.totensor(i, l)
            random.seed(seed); np.random.seed(seed)
            prm = do.draw_train_params(lab, (64, 64), 255)
            io, lo = do.prepare_tile(img, lab, (64, 64), *prm)
            assert np.array_equal(it.numpy(), io) and np.array_equal(lt.numpy(), lo), 'g17 %s %d' % (tag, rep)
            out['%s_%d_img' % (tag, rep)] = it.numpy()[:, ::3, ::3]
            out['%s_%d_lbl' % (tag, rep)] = lt.numpy().astype(np.uint8)
            out['%s_%d_prm' % (tag, rep)] = np.array([prm[0], prm[1], int(prm[2]), prm[3]], dtype=np.int32)
    # non-trivial mean / std (BaseDataset defaults, base_dataset.py:9) through normalize + totensor only
    ds = base.BaseDataset(mode='val', crop_size=(64, 64))
    img = (fm.uniform01('g17/norm/img', 40 * 48 * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(40, 48, 3).numpy()
    it = ds.totensor(ds.normalize(img))
    io, _ = do.prepare_tile(img, None, (40, 48), 0, 0, False, 0, mean=ds.mean, std=ds.std)
    assert np.array_equal(it.numpy(), io), 'g17 normalize'
    out['norm_img'] = it.numpy()
    # label re-indexing: GFSSegVal.__getitem__ (oem.py:98-137) on a tile holding every class, 0 and 255
    lab = (fm.uniform01('g17/remap/lab', 32 * 32) * 13).floor().to(torch.uint8).reshape(32, 32).numpy()
    lab[lab == 12] = 255
    rgb = np.zeros((3, 32, 32), dtype=np.uint8)
    arrays.update({'R/images/t.tif': rgb, 'R/labels/t.tif': lab[None]})
    lst = os.path.join('/tmp', 'g17_val_list.txt')
    open(lst, 'w').write('t\n')
    real_exists = os.path.exists
    os.path.exists = lambda p: True if p == 'R/labels/t.tif' else real_exists(p)
    try:
        for ub, un in ((True, True), (True, False), (False, True)):
            dv = oem.GFSSegVal('R', lst, 0, base_size=(32, 32), resize_label=False, use_novel=un, use_base=ub)
            _, lt, _ = dv[0]
            lut = do.remap_lut(dv.base_classes, dv.novel_classes, use_base=ub, use_novel=un)
            assert np.array_equal(lt.numpy(), lut[lab].astype(np.int64)), 'g17 remap'
            out['remap_%d%d' % (ub, un)] = lt.numpy().astype(np.uint8)
    finally:
        os.path.exists = real_exists
    save('g17_oem_tiles', **out)


# ------------------------------------------------------------------------------------------ G18: probability-map fusion
def g18():
    """fusemat.py executed as the script it is (placeholders of its fusion_list / output_path replaced by temporary directories holding three
    models' .mat dumps of two tiles); its PNGs (nearest-neighbour x16) give back the fused 64 x 64 label maps, which the numpy restatement must equal."""
    import scipy.io
    import tempfile
    from PIL import Image
    from oracle import data_oracle as do
    tmp = tempfile.mkdtemp(prefix='g18_')
    dirs = [os.path.join(tmp, 'm%d' % m) for m in range(3)]
    maps = {}
    for m, d in enumerate(dirs):
        os.makedirs(d)
        for tile in ('a', 'b'):
            arr = fm.sym('g18/m%d/%s' % (m, tile), (1, 8, 64, 64), 3.0).numpy()
            if tile == 'b':
                arr[:, :, :8] = np.round(arr[:, :, :8])          # exact ties between classes: first maximum must win
            scipy.io.savemat(os.path.join(d, tile + '.mat'), {'outputs': arr})
            maps.setdefault(tile, []).append(arr[0])
    out_dir = os.path.join(tmp, 'out')
    src = open(os.path.join(REF, 'fusemat.py')).read()
    src = src.replace("'PATH_OF_PROBABILITY_MAPS_FOR_FUSION_1'", repr(dirs[0])).replace("'PATH_OF_PROBABILITY_MAPS_FOR_FUSION_2'", repr(dirs[1]))
    src = src.replace("'PATH_OF_PROBABILITY_MAPS_FOR_FUSION_3',\n        '...'", repr(dirs[2])).replace("'PATH_OF_OUTPUT_PROBABILITY_MAPS'", repr(out_dir))
    import scipy as _scipy
    exec(compile(src, 'fusemat.py', 'exec'), {'__name__': '__main__'})
    res = {}
    for tile in ('a', 'b'):
        png = np.array(Image.open(os.path.join(out_dir, tile + '.png')))
        assert png.shape == (1024, 1024)
        lab = png[::16, ::16]
        assert np.array_equal(np.repeat(np.repeat(lab, 16, 0), 16, 1), png)
        assert np.array_equal(lab, do.fuse_probability_maps(maps[tile])), 'g18 ' + tile
        res['fused_' + tile] = lab.astype(np.uint8)
    save('g18_fusion', **res)



# ------------------------------------------------------------------------------------------ G19: fine-tune pair reader
def g19():
    """dataset/oem_ft.py GFSSegTrain as the reference runs it in ft_pop.py:157-160: class -> id lists (_filter_and_map_ids, written to and re-read
    from train_base_class<c>.txt), the support / base lists (_get_supp_list, update_base_list) and (novel, base) training pairs
    (_get_train_sample) under fixed seeds, on synthetic tiles fed through the same rasterio stand-in and numpy copyMakeBorder as G17."""
    import importlib.util
    import random
    import shutil
    import tempfile
    from oracle import data_oracle as do

    def copy_make_border(src, top, bottom, left, right, border_type, value=0):
        v = value[0] if isinstance(value, (tuple, list)) else value
        pads = ((top, bottom), (left, right)) + (((0, 0),) if src.ndim == 3 else ())
        return np.pad(src, pads, constant_values=v)
    sys.modules['cv2'].copyMakeBorder = copy_make_border
    sys.modules['cv2'].BORDER_CONSTANT = 0
    ids, imgs, labs = do.ft_tiles()
    arrays = {}
    for id_ in ids:
        arrays['R/images/%s.tif' % id_] = np.ascontiguousarray(np.rollaxis(imgs[id_], 2, 0))      # rasterio returns [C,H,W]
        arrays['R/labels/%s.tif' % id_] = labs[id_][None]
    rio = types.ModuleType('rasterio')
    rio.open = lambda path: types.SimpleNamespace(read=lambda: arrays[path])
    sys.modules['rasterio'] = rio
    pkg = types.ModuleType('ref_dataset19'); pkg.__path__ = [os.path.join(REF, 'dataset')]
    sys.modules['ref_dataset19'] = pkg

    def load(name):
        spec = importlib.util.spec_from_file_location('ref_dataset19.' + name, os.path.join(REF, 'dataset', name + '.py'), submodule_search_locations=None)
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = 'ref_dataset19'
        sys.modules['ref_dataset19.' + name] = mod
        spec.loader.exec_module(mod)
        return mod
    load('base_dataset')
    oem_ft = load('oem_ft')
    shot, seed, crop = 2, 123, (64, 64)
    base_classes, novel_classes = set(range(1, 8)), set(range(8, 12))
    read_label = lambda i: labs[i]          # noqa: E731
    read_image = lambda i: imgs[i]          # noqa: E731
    out = {}
    for filt in (False, True):
        tmp = tempfile.mkdtemp(prefix='g19_')
        list_dir = os.path.join(tmp, 'list')
        for d in (list_dir, list_dir + '_filter'):
            os.makedirs(d)
        lst = os.path.join(list_dir, 'train.txt')
        open(lst, 'w').write(''.join(i + '\n' for i in ids))
        b2i, n2i = do.filter_and_map_ids(ids, read_label, base_classes, novel_classes, filter_intersection=False)
        novel_ids = []
        for c in sorted(novel_classes):                      # all_<shot>shot_seed<seed>.txt: `shot` support tiles per novel class (an input of the reader)
            novel_ids += n2i[c][:shot]
        for d in (list_dir, list_dir + '_filter'):
            open(os.path.join(d, 'all_%dshot_seed%d.txt' % (shot, seed)), 'w').write(''.join(i + '\n' for i in novel_ids))
        tag = 'f%d' % int(filt)
        random.seed(7); np.random.seed(7)
        ds = oem_ft.GFSSegTrain('R', lst, 0, shot=shot, mode='train', crop_size=crop, base_size=(64, 64), seed=seed, filter=filt)
        ds.mean, ds.std = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]          # the values oem.py:26-27 trains with (oem_ft.py keeps BaseDataset's ImageNet defaults -- see below)
        random.seed(7); np.random.seed(7)
        b2i_o, _ = do.filter_and_map_ids(ids, read_label, base_classes, novel_classes, filter_intersection=filt)
        base_o = do.sample_base_ids(b2i_o, base_classes, shot)
        for c in base_classes:
            assert list(ds.base_cls_to_ids[c]) == list(b2i_o[c]), ('g19 class map', c)
            out['%s_cls%d' % (tag, c)] = np.array(b2i_o[c] or [''], dtype='U8')
        assert ds.base_id_list == base_o and ds.supp_cls_id_list == novel_ids + base_o, 'g19 supp list'
        assert len(ds) == len(base_o)
        out[tag + '_base0'] = np.array(base_o, dtype='U8')
        # a second construction re-reads the class lists from the files the first one wrote
        random.seed(11)
        ds2 = oem_ft.GFSSegTrain('R', lst, 0, shot=shot, mode='train', crop_size=crop, base_size=(64, 64), seed=seed, filter=filt)
        random.seed(11)
        assert ds2.base_id_list == do.sample_base_ids(b2i_o, base_classes, shot), 'g19 re-read'
        # pairs, then update_base_list, then more pairs -- one RNG stream throughout
        random.seed(21); np.random.seed(21)
        got = [ds[i] for i in (0, 5, len(ds) - 1)]
        ds.update_base_list()
        got += [ds[i] for i in (1, 2)]
        random.seed(21); np.random.seed(21)
        bl = base_o
        exp = [do.ft_pair(i, bl, novel_ids, read_image, read_label, crop) for i in (0, 5, len(bl) - 1)]
        bl = do.sample_base_ids(b2i_o, base_classes, shot)
        assert ds.base_id_list == bl, 'g19 update_base_list'
        exp += [do.ft_pair(i, bl, novel_ids, read_image, read_label, crop) for i in (1, 2)]
        out[tag + '_base1'] = np.array(bl, dtype='U8')
        for k, (g, e) in enumerate(zip(got, exp)):
            assert np.array_equal(g[0].numpy(), e[0]) and np.array_equal(g[1].numpy(), e[1]) and np.array_equal(g[2].numpy(), e[2]) \
                and np.array_equal(g[3].numpy(), e[3]) and g[4] == e[4], 'g19 pair %d' % k
            out['%s_p%d_img' % (tag, k)] = g[0].numpy()[:, ::4, ::4]
            out['%s_p%d_lbl' % (tag, k)] = g[1].numpy().astype(np.uint8)
            out['%s_p%d_imgb' % (tag, k)] = g[2].numpy()[:, ::4, ::4]
            out['%s_p%d_lblb' % (tag, k)] = g[3].numpy().astype(np.uint8)
            out['%s_p%d_id' % (tag, k)] = np.array([g[4]], dtype='U8')
            out['%s_p%d_prm' % (tag, k)] = np.array([list(map(int, e[5])), list(map(int, e[6]))], dtype=np.int32)
        shutil.rmtree(tmp)
    out['novel_ids'] = np.array(novel_ids, dtype='U8')
    # oem_ft.py does NOT override BaseDataset's ImageNet mean / std (oem.py:26-27 does): a fine-tune pair is normalised differently from a
    # base-training tile.  Reproduced on purpose; one pair with the reader's own defaults:
    tmp = tempfile.mkdtemp(prefix='g19_')
    os.makedirs(os.path.join(tmp, 'list'))
    lst = os.path.join(tmp, 'list', 'train.txt')
    open(lst, 'w').write(''.join(i + '\n' for i in ids))
    open(os.path.join(tmp, 'list', 'all_%dshot_seed%d.txt' % (shot, seed)), 'w').write(''.join(i + '\n' for i in novel_ids))
    random.seed(5); np.random.seed(5)
    ds = oem_ft.GFSSegTrain('R', lst, 0, shot=shot, mode='train', crop_size=crop, base_size=(64, 64), seed=seed)
    g = ds[3]
    out['default_mean'], out['default_std'] = np.array(ds.mean), np.array(ds.std)
    out['default_img'], out['default_imgb'] = g[0].numpy()[:, ::4, ::4], g[2].numpy()[:, ::4, ::4]
    out['default_base'] = np.array(ds.base_id_list, dtype='U8')
    shutil.rmtree(tmp)
    save('g19_oem_ft', **out)


ALL = dict(g19=g19, g18=g18, g17=g17, g13=g13, g14=g14, g15=g15, g16=g16, g12=g12, g11=g11, g1=g1, g2=g2, g3=g3, g3b=g3b, g4=g4, g5=g5, g6=g6, g7=g7, g8=g8, g9=g9, g10=g10)

if __name__ == '__main__':
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print('==', w)
        ALL[w]()
