"""Pins oracle/pop_oracle.py to the golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only.  Bit-exact where the op sequence is identical
(torch.equal held against the reference at generation time; across machines we allow a few ULP)."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import formula as fm
from oracle import pop_oracle as po

TOL = dict(rtol=2e-5, atol=2e-6)


def close(a, b, **kw):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    tol = dict(TOL); tol.update(kw)
    np.testing.assert_allclose(a, b, **tol)


def build(is_ft=False, n_novel=0):
    m = po.PopOracle(n_base=7, criterion=po.OrthLossOracle(255), is_ft=is_ft, n_novel=n_novel, backbone='resnet50')
    return fm.load_formula_weights(m)


def test_formula_is_stable():
    # known answers so that a silent change of the generator cannot re-pin everything
    u = fm.uniform01('backbone.conv1.weight', 4)
    assert u.dtype == torch.float64 and float(u.min()) >= 0 and float(u.max()) < 1
    a = fm.sym('x', (3,), 1.0)
    b = fm.sym('x', (3,), 1.0)
    assert torch.equal(a, b)
    g = golden('g1_decompose')
    assert g['proj'].shape == (2, 7, 24)


def test_g1_decompose():
    g = golden('g1_decompose')
    feats = fm.sym('g1/feats', (2, 512, 24), 1.0)
    bb, bn = fm.sym('g1/bb', (1, 7, 512), 1.0), fm.sym('g1/bn', (1, 4, 512), 1.0)
    fg, bg = po.orthogonal_decompose(feats, bb)
    close(fg[:, :, ::32], g['fg_sub']); close(bg, g['bg'])
    _, fgn, bg2 = po.orthogonal_decompose(feats, bb, bn)
    close(fgn[:, :, ::32], g['fgn_sub']); close(bg2, g['bg2'])
    # properties: residual is orthogonal to every (non-orthogonal-set) basis only after exact projection
    s = F.normalize(bb, dim=-1)
    close(torch.matmul(s, feats), g['proj'])


def test_g2_head():
    g = golden('g2_head')
    m = build()
    feats = fm.sym('g2/feats', (2, 512, 8, 8), 1.0).requires_grad_(True)
    coef = fm.sym('g2/coef', (2, 8, 8, 8), 1.0)
    preds = po.head_base(m, feats)
    (preds * coef).sum().backward()
    close(preds, g['preds']); close(feats.grad, g['dfeats'], atol=2e-5)
    close(m.base_emb.grad, g['d_base_emb'], atol=2e-5)
    close(m.classifier[0].weight.grad[::8, ::8, 0, 0], g['d_cls0'], atol=2e-5)
    close(m.classifier[4].weight.grad[0, :, 0, 0], g['d_cls4'], atol=2e-5)
    m2 = build(True, 4).eval()
    close(po.head_all(m2, feats.detach())[0], golden('g2_head_all')['preds'])


def test_head_collapse_identity():
    """for a foreground class pred = a_k*max(p,0) + b_k*max(-p,0) with a_k = MLP(s_k), b_k = MLP(-s_k).
    The HIP head uses this identity; pin it against the direct form here."""
    m = build()
    feats = fm.sym('g2/feats', (2, 512, 8, 8), 1.0)
    preds = po.head_base(m, feats)
    s = F.normalize(m.base_emb.detach(), dim=-1)                      # [K,C]
    p = torch.einsum('kc,bcn->bkn', s, feats.flatten(2))              # [B,K,N]
    rows = torch.cat([s, -s], 0).view(14, 512, 1, 1)
    ab = po.classifier_forward(m.classifier, rows).view(2, 7)
    fg = ab[0].view(1, 7, 1) * p.clamp(min=0) + ab[1].view(1, 7, 1) * (-p).clamp(min=0)
    close(fg.view(2, 7, 8, 8), preds[:, 1:].detach().numpy(), rtol=1e-4, atol=1e-5)


def test_g3_loss():
    g = golden('g3_loss')
    co = po.OrthLossOracle(255)
    preds = fm.sym('g3/preds', (2, 8, 8, 8), 2.0).requires_grad_(True)
    target = fm.formula_mask(2, 64, 64, 8, tag='g3/mask', block=8, ignore_rows=5)
    e = F.normalize(fm.sym('g3/emb', (7, 512), 1.0), dim=-1)
    sim = (e @ e.t()).requires_grad_(True)
    d = co(preds, target, proto_sim=sim)
    d['total_loss'].backward()
    close(d['total_loss'], g['total']); close(d['seg_loss'], g['seg']); close(d['orth_loss'], g['orth'])
    close(preds.grad, g['dpreds']); close(sim.grad, g['dsim'])
    rect = fm.sym('g3/rect', (4, 11), 1.0).requires_grad_(True)
    o = co.get_orth_loss(rect); o.backward()
    close(o, g['orth_rect']); close(rect.grad, g['d_rect'])
    preds12 = fm.sym('g3/preds12', (2, 12, 8, 8), 2.0).requires_grad_(True)
    t12 = fm.formula_mask(2, 64, 64, 12, tag='g3/mask12', block=8, ignore_rows=3)
    d12 = co(preds12, t12, is_ft=True, proto_sim=rect.detach())
    d12['total_loss'].backward()
    close(d12['total_loss'], g['total12']); close(d12['seg_loss'], g['seg12']); close(preds12.grad, g['dpreds12'])


@pytest.mark.parametrize('tag,feat,outf,hw', [('a', 64, 64, 12), ('b', 128, 64, 16)])
def test_g4_ppm(tag, feat, outf, hw):
    g = golden('g4_ppm_' + tag)
    ora = po.make_ppm(feat, outf)
    ora.load_state_dict({k: fm.formula_tensor('g4' + tag + '/' + k, v) for k, v in ora.state_dict().items()})
    x = fm.sym('g4%s/x' % tag, (2, feat, hw, hw), 1.0).relu_().requires_grad_(True)
    coef = fm.sym('g4%s/coef' % tag, (2, outf, hw, hw), 1.0)
    ora.train()
    y = po.ppm_forward(ora, x); (y * coef).sum().backward()
    close(y, g['y'], atol=1e-5); close(x.grad, g['dx'], atol=1e-5)
    close(ora.bottleneck[0].weight.grad[::4, ::16], g['d_bott_w'], atol=1e-5)
    close(ora.stages[3][2].running_mean, g['rm_stage3']); close(ora.bottleneck[1].running_var, g['rv_bott'])
    ora.eval()
    close(po.ppm_forward(ora, x.detach()), g['y_eval'], atol=1e-5)


G5_CASES = {'s1_ds': (64, 64, 1, 1, True), 's1_id': (256, 64, 1, 1, False), 's2_ds': (256, 128, 2, 1, True),
            'd2_ds': (512, 256, 1, 2, True), 'd4_id': (1024, 256, 1, 4, False)}


@pytest.mark.parametrize('name', list(G5_CASES))
def test_g5_bottleneck(name):
    inp, pl, st, dil, ds = G5_CASES[name]
    g = golden('g5_bottleneck_' + name)
    ora = po.make_bottleneck(inp, pl, st, dil, ds)
    ora.load_state_dict({k: fm.formula_tensor('g5' + name + '/' + k, v) for k, v in ora.state_dict().items()})
    x = fm.sym('g5%s/x' % name, (2, inp, 16, 16), 1.0).relu_().requires_grad_(True)
    ora.train()
    y = po.bottleneck_forward(ora, x)
    coef = fm.sym('g5%s/coef' % name, tuple(y.shape), 1.0)
    (y * coef).sum().backward()
    close(y[:, ::4], g['y'], atol=1e-5); close(x.grad[:, ::4], g['dx'], atol=1e-5)
    close(ora.conv2.weight.grad[::4, ::4], g['d_conv2_w'], atol=1e-5)
    close(ora.bn3.weight.grad, g['d_bn3_gamma'], atol=1e-4)
    close(ora.bn2.running_mean, g['rm_bn2']); close(ora.bn2.running_var, g['rv_bn2'])
    ora.eval()
    close(po.bottleneck_forward(ora, x.detach())[:, ::4], g['y_eval'], atol=1e-5)


def test_g9_masked_average_pooling():
    feat = fm.sym('g9/feat', (2, 32, 8, 8), 1.0)
    m = (fm.uniform01('g9/mask', 2 * 64 * 64).reshape(2, 1, 64, 64) > 0.5).float()
    close(po.masked_average_pooling(feat, m), golden('g9_map')['proto'])


def test_g10_iou():
    g = golden('g10_iou')
    pred = (fm.uniform01('g10/pred', 2 * 64 * 64) * 8).floor().long().reshape(2, 64, 64)
    tgt = (fm.uniform01('g10/tgt', 2 * 64 * 64) * 8).floor().long().reshape(2, 64, 64)
    tgt[0, :5] = 255
    i, u, t = po.intersection_and_union(pred, tgt, 8, 255)
    assert np.array_equal(i.numpy(), g['inter']) and np.array_equal(u.numpy(), g['union']) and np.array_equal(t.numpy(), g['target'])


G12 = {'a': dict(dilated=True, os=8, multi_grid=True, relu_l3=True, relu_l4=False),
       'b': dict(dilated=True, os=16, multi_grid=True, relu_l3=False, relu_l4=False)}


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_g12_constructor_kwargs(tag):
    """multi_grid / relu_l3 / relu_l4 (networks/backbones/resnet.py:81-121) through the oracle against the reference's vectors."""
    g = golden('g12_kwargs_' + tag)
    o = fm.load_formula_weights(po.PopOracle(n_base=7, criterion=po.OrthLossOracle(255), backbone='resnet50', **G12[tag])).train()
    img = fm.formula_image(2, 96, 128, 'g12%s/img' % tag)
    mask = fm.formula_mask(2, 96, 128, 8, 'g12%s/mask' % tag, ignore_rows=5)
    d = o(img, mask)
    d['total_loss'].backward()
    close(d['total_loss'].detach(), g['total']); close(d['seg_loss'].detach(), g['seg']); close(d['orth_loss'].detach(), g['orth'])
    close(o.base_emb.grad, g['d_base_emb'], rtol=1e-4, atol=1e-7)
    close(o.backbone.layer4[2].bn3.running_mean, g['rm_l4'])
    o.eval()
    with torch.no_grad():
        close(o(img), g['logits_eval'], rtol=1e-4, atol=1e-6)


def test_g11_eval_confusion():
    """eval_base.py:166-199 / eval_ft.py:166-181: prediction mask, confusion matrix and IoU vector."""
    logits = fm.sym('g11/logits', (2, 12, 16, 12), 2.0)
    label = (fm.uniform01('g11/label', 2 * 128 * 96) * 12).floor().long().reshape(2, 128, 96)
    label[1, :9] = 255
    for tag, pad in (('plain', False), ('ft', True)):
        g = golden('g11_eval_' + tag)
        pred, cm = po.eval_confusion(logits, label, 12, 255, pad_to_longside=pad)
        assert np.array_equal(pred, g['pred']) and np.array_equal(cm, g['cm'])
        iou, b, n, t = po.miou_from_confusion(cm, 7)
        assert np.allclose(iou, g['iou'], equal_nan=True) and np.allclose([b, n, t], g['miou'])


def test_index_rules_match_aten():
    # adaptive-avg-pool bins and bilinear taps (both align modes) restated in closed form
    for n, s in [(64, 1), (64, 2), (64, 3), (64, 6), (12, 6), (12, 3), (16, 6), (32, 3)]:
        x = torch.arange(n, dtype=torch.float32).view(1, 1, n, 1).expand(1, 1, n, n).contiguous()
        ref = F.adaptive_avg_pool2d(x, (s, s))[0, 0, :, 0]
        mine = torch.tensor([float(sum(range(a, b))) / (b - a) for a, b in po.adaptive_bins(n, s)])
        close(mine, ref.numpy())
    for (i, o, ac) in [(1, 64, False), (2, 64, False), (3, 64, False), (6, 64, False), (6, 12, False), (64, 512, True), (8, 64, True)]:
        v = fm.sym('taps%d_%d' % (i, o), (i,), 1.0)
        ref = F.interpolate(v.view(1, 1, i, 1).expand(1, 1, i, 2).contiguous(), size=(o, 2), mode='bilinear', align_corners=ac)[0, 0, :, 0]
        mine = torch.tensor([(1 - l) * float(v[a]) + l * float(v[b]) for a, b, l in po.bilinear_taps(i, o, ac)])
        close(mine, ref.numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------- full-network pins
@pytest.mark.slow
def test_g6_full_r50():
    g = golden('g6_full_r50')
    m = build().train()
    img = fm.formula_image(2, 512, 512, 'g6/img')
    mask = fm.formula_mask(2, 512, 512, 8, 'g6/mask')
    feats = po.features_of(m, img)
    logits = po.head_base(m, feats)
    e = F.normalize(m.base_emb, dim=-1)
    d = m.criterion(logits, mask, proto_sim=e @ e.t())
    d['total_loss'].backward()
    close(logits, g['logits'], rtol=1e-4, atol=1e-5)
    close(d['total_loss'], g['total']); close(d['seg_loss'], g['seg']); close(d['orth_loss'], g['orth'], atol=1e-7)
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1e30)
    close(gn, g['gnorm'], rtol=1e-4)
    close(m.base_emb.grad, g['d_base_emb'], rtol=1e-3, atol=1e-6)
    close(m.backbone.conv1.weight.grad, g['d_conv1'], rtol=1e-3, atol=1e-6)
    close(m.backbone.bn1.running_mean, g['rm_bn1']); close(m.backbone.bn1.running_var, g['rv_bn1'])
    up = F.interpolate(logits.detach(), size=(512, 512), mode='bilinear', align_corners=True)
    am = up.argmax(1).numpy().astype(np.uint8)
    top2 = up.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    bad = (am != g['argmax']) & (margin > 1e-5)      # bit-exact except numerically tied pixels
    assert bad.sum() == 0, 'argmax differs on %d non-tied pixels' % bad.sum()
    m.eval()
    close(m(img), g['logits_eval'], rtol=1e-4, atol=1e-5)


@pytest.mark.slow
def test_g7_ft():
    g = golden('g7_ft')
    m = build(True, 4)
    po.init_cls_n(m)
    with torch.no_grad():
        for k, p in m.classifier_n.named_parameters():
            p.add_(fm.sym('g7/cn/' + k, tuple(p.shape), 0.01))
    img = fm.formula_image(1, 512, 512, 'g7/img'); img_b = fm.formula_image(1, 512, 512, 'g7/img_b')
    mask = fm.formula_mask(1, 512, 512, 4, 'g7/mask', ignore_rows=0, lo=8); mask[mask == 8] = 255
    mask_b = fm.formula_mask(1, 512, 512, 8, 'g7/mask_b', ignore_rows=0)
    po.train_mode(m)
    d = m(img, mask, img_b, mask_b)
    d['total_loss'].backward()
    close(d['total_loss'], g['total']); close(d['seg_loss'], g['seg']); close(d['orth_loss'], g['orth'])
    assert np.array_equal(mask_b.numpy().astype(np.uint8), g['mask_b_new'])
    close(m.novel_emb.grad, g['d_novel_emb'], rtol=1e-3, atol=1e-6)
    m.eval()
    close(m(img), g['preds_all'], rtol=1e-4, atol=1e-5)


@pytest.mark.slow
def test_g8_trajectory_and_param_groups():
    g = golden('g8_traj')
    m = build().train()
    groups, keys = po.param_groups(m, lr=1e-5)
    sizes = [(len(x['params']), sum(p.numel() for p in x['params'])) for x in groups]
    assert np.array_equal(np.array(sizes), g['group_sizes'])
    assert sizes == [(159, 23508032), (6, 3072), (15, 23861760)]        # SURVEY 8 a-12 [probe]
    assert list(g['group_keys1']) == keys[1]
    opt = torch.optim.AdamW(groups, lr=1e-5, weight_decay=1e-4)
    img = fm.formula_image(2, 512, 512, 'g6/img'); mask = fm.formula_mask(2, 512, 512, 8, 'g6/mask')
    for step in range(3):
        losses, norm = po.train_step(m, opt, img, mask, clip_grad=5.0, double_step=True)
        close([losses['total_loss'], losses['seg_loss'], losses['orth_loss']], g['losses'][step], rtol=2e-3, atol=1e-5)
        close(norm, g['norms'][step], rtol=5e-3)


def test_g17_tile_preparation_oracle():
    """Row f-2: the numpy restatement of dataset/base_dataset.py / oem.py against the golden produced by the reference's own code (same seeds
    -> same crops), plus the host-side draw / lookup-table helpers of the product against the oracle's."""
    import random
    from oracle import data_oracle as do
    from segland_amd.dataset import augment as aug
    g = golden('g17_oem_tiles')
    for tag, (H, W) in (('small', (50, 70)), ('exact', (64, 64)), ('large', (100, 90))):
        img = (fm.uniform01('g17/%s/img' % tag, H * W * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(H, W, 3).numpy()
        lab = (fm.uniform01('g17/%s/lab' % tag, H * W) * 12).floor().to(torch.uint8).reshape(H, W).numpy()
        lab[:7] = 255
        for rep in range(3):
            seed = 100 * rep + H
            random.seed(seed); np.random.seed(seed)
            prm = do.draw_train_params(lab, (64, 64), 255)
            random.seed(seed); np.random.seed(seed)
            assert aug.draw_train_params(lab, (64, 64), 255) == prm
            assert [prm[0], prm[1], int(prm[2]), prm[3]] == g['%s_%d_prm' % (tag, rep)].tolist()
            io, lo = do.prepare_tile(img, lab, (64, 64), *prm)
            assert np.array_equal(io[:, ::3, ::3], g['%s_%d_img' % (tag, rep)]) and np.array_equal(lo.astype(np.uint8), g['%s_%d_lbl' % (tag, rep)])
    for ub, un in ((True, True), (True, False), (False, True)):
        lut = do.remap_lut(set(range(1, 8)), set(range(8, 12)), ub, un)
        assert np.array_equal(lut, aug.remap_lut(set(range(1, 8)), set(range(8, 12)), ub, un))
        lab = (fm.uniform01('g17/remap/lab', 32 * 32) * 13).floor().to(torch.uint8).reshape(32, 32).numpy()
        lab[lab == 12] = 255
        assert np.array_equal(lut[lab], g['remap_%d%d' % (ub, un)])


def test_g18_fusion_oracle():
    """Row f-3: numpy restatement of fusemat.py:35-52 against the label maps the reference script itself produced (golden G18)."""
    from oracle import data_oracle as do
    g = golden('g18_fusion')
    for tile in ('a', 'b'):
        maps = []
        for m in range(3):
            arr = fm.sym('g18/m%d/%s' % (m, tile), (1, 8, 64, 64), 3.0).numpy()
            if tile == 'b':
                arr[:, :, :8] = np.round(arr[:, :, :8])
            maps.append(arr[0])
        assert np.array_equal(do.fuse_probability_maps(maps), g['fused_' + tile])


@pytest.mark.parametrize('filt', [False, True])
def test_g19_pair_reader_lists_and_draws(tmp_path, filt):
    """Row f-2, the fine-tune pair reader: golden G19 is what dataset/oem_ft.py of the reference produced (class -> id lists, support / base
    lists before and after update_base_list, (novel, base) pairs with their draws) under fixed seeds.  The numpy oracle AND the product's host
    logic (segland_amd/dataset/oem_ft.py: lists, draws, the 0 -> ignore rule before the crop draw) must reproduce it from the same seeds."""
    import random
    from oracle import data_oracle as do
    from g19_common import product_reader
    g = golden('g19_oem_ft')
    tag = 'f%d' % int(filt)
    novel_ids = g['novel_ids'].tolist()
    ids, imgs, labs = do.ft_tiles()
    base_classes, novel_classes = set(range(1, 8)), set(range(8, 12))
    # oracle
    b2i, _ = do.filter_and_map_ids(ids, lambda i: labs[i], base_classes, novel_classes, filter_intersection=filt)
    for c in base_classes:
        assert (b2i[c] or ['']) == g['%s_cls%d' % (tag, c)].tolist()
    random.seed(7); np.random.seed(7)
    base0 = do.sample_base_ids(b2i, base_classes, 2)
    assert base0 == g[tag + '_base0'].tolist()
    # product: same seeds, same lists (first construction scans the labels, the second re-reads the class files it wrote)
    Reader = product_reader(tmp_path, filt, novel_ids=novel_ids)
    random.seed(7); np.random.seed(7)
    ds = Reader()
    assert ds.base_id_list == base0 and ds.supp_cls_id_list == novel_ids + base0 and len(ds) == len(base0)
    for c in base_classes:
        assert (list(ds.base_cls_to_ids[c]) or ['']) == g['%s_cls%d' % (tag, c)].tolist()
    random.seed(7); np.random.seed(7)
    assert Reader().base_id_list == base0
    # pairs: product draws == oracle draws == the reference's
    random.seed(21); np.random.seed(21)
    got = [ds[i] for i in (0, 5, len(ds) - 1)]
    ds.update_base_list()
    assert ds.base_id_list == g[tag + '_base1'].tolist()
    got += [ds[i] for i in (1, 2)]
    random.seed(21); np.random.seed(21)
    bl = base0
    exp = [do.ft_pair(i, bl, novel_ids, lambda i: imgs[i], lambda i: labs[i], (64, 64)) for i in (0, 5, len(bl) - 1)]
    bl = do.sample_base_ids(b2i, base_classes, 2)
    exp += [do.ft_pair(i, bl, novel_ids, lambda i: imgs[i], lambda i: labs[i], (64, 64)) for i in (1, 2)]
    for k, (s, e) in enumerate(zip(got, exp)):
        (tile, tile_b), (prm, prm_b), id_ = s
        want = g['%s_p%d_prm' % (tag, k)].tolist()
        assert id_ == e[4] == g['%s_p%d_id' % (tag, k)][0]
        assert [list(map(int, prm)), list(map(int, prm_b))] == want == [list(map(int, e[5])), list(map(int, e[6]))]
        assert 0 not in np.unique(tile[1])                      # oem_ft.py:197: unlabeled pixels of the novel tile are ignore
        assert np.array_equal(e[0][:, ::4, ::4], g['%s_p%d_img' % (tag, k)]) and np.array_equal(e[1].astype(np.uint8), g['%s_p%d_lbl' % (tag, k)])
        assert np.array_equal(e[2][:, ::4, ::4], g['%s_p%d_imgb' % (tag, k)]) and np.array_equal(e[3].astype(np.uint8), g['%s_p%d_lblb' % (tag, k)])
        # the product's raw tiles + draws through the oracle's pixel path give the golden pixels too (the GPU test does this with the kernel)
        io, lo = do.prepare_tile(tile[0], tile[1], (64, 64), *prm)
        assert np.array_equal(io[:, ::4, ::4], g['%s_p%d_img' % (tag, k)]) and np.array_equal(lo.astype(np.uint8), g['%s_p%d_lbl' % (tag, k)])


def test_pair_reader_rejects_what_the_reference_cannot_run(tmp_path):
    from g19_common import product_reader
    from segland_amd.dataset.oem_ft import MEAN, STD, PairReader
    g = golden('g19_oem_ft')
    assert np.allclose(g['default_mean'], MEAN) and np.allclose(g['default_std'], STD)       # oem_ft.py keeps BaseDataset's ImageNet statistics
    r = PairReader()
    with pytest.raises(RuntimeError, match='val_supp'):
        r._init_lists('x/list/train.txt', 1, 'val_supp', (64, 64), 255, 123, False, True)
    Reader = product_reader(tmp_path, False, novel_ids=g['novel_ids'].tolist())

    class NoBase(Reader):
        def __init__(self):
            self._init_lists(os.path.join(str(tmp_path), 'list', 'train.txt'), 2, 'train', (64, 64), 255, 123, False, False)
    with pytest.raises(RuntimeError, match='use_base=False'):
        NoBase()


def test_g3b_loss_with_aux_preds():
    """criterion.py:56-60 (the aux_preds branch): the oracle against what the reference itself produced (tests/golden/make_golden.py g3b)."""
    g = golden('g3b_loss_aux')
    co = po.OrthLossOracle(255)
    preds = fm.sym('g3b/preds', (2, 8, 8, 8), 2.0).requires_grad_(True)
    aux = fm.sym('g3b/aux', (2, 8, 16, 16), 1.5).requires_grad_(True)
    target = fm.formula_mask(2, 64, 64, 8, tag='g3b/mask', block=8, ignore_rows=7)
    e = F.normalize(fm.sym('g3b/emb', (7, 512), 1.0), dim=-1)
    sim = (e @ e.t()).requires_grad_(True)
    d = co(preds, target, proto_sim=sim, aux_preds=aux)
    assert sorted(d) == ['aux_loss', 'orth_loss', 'seg_loss', 'total_loss']
    d['total_loss'].backward()
    close(d['total_loss'], g['total']); close(d['seg_loss'], g['seg']); close(d['aux_loss'], g['aux']); close(d['orth_loss'], g['orth'])
    close(preds.grad, g['dpreds']); close(aux.grad, g['daux']); close(sim.grad, g['dsim'])
