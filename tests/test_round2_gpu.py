"""Round-2 parity gates (VERDICT r1 "Next round" item 1 + ADVICE r1): the bench dtype in TRAIN mode, ResNet-101 in bf16 at the bench
size, the product under RCCL DDP, argmax / pseudo-labels against the same-box oracle with a 1e-5 margin, golden G1 on the GPU, and the
regressions the advisor named (stale eval coefficients after a frozen-affine train forward, pin-memory DataLoader threads next to a
HIP-graph capture, AdamW with per-parameter step counts, per-GPU batch 1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import golden
from oracle import formula as fm

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(backbone='resnet50', dtype=torch.bfloat16, is_ft=False, criterion=True, **kw):
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    return GFSS_Model(n_base=7, criterion=OrthLoss(255) if criterion else None, backbone=backbone, pretrained_model=None, dilated=True, os=8,
                      is_ft=is_ft, n_novel=4 if is_ft else 0, compute_dtype=dtype, **kw)


def _batch(B, size, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    img = torch.randn(B, 3, size, size, generator=g)
    mask = torch.randint(0, 8, (B, size, size), generator=g, dtype=torch.int64)
    mask[0, :size // 10] = 255
    return img, mask


def _round_weights_to_bf16_(model):
    """Conv weights as the bf16 kernels see them (fp32 masters rounded to bf16): the oracle then differs from the HIP path only by the
    activation rounding, not by the weight rounding."""
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Conv2d) and m.weight.shape[1] >= 32:
                m.weight.copy_(m.weight.to(torch.bfloat16).float())


# --------------------------------------------------------------------------------------------- (e) golden G1 on the GPU
def test_g1_decompose_gpu(hip):
    """pop_decompose_fwd (pspnet_pop.py:95-121): projections and the background residual against golden G1, single and dual basis."""
    from segland_amd import ops
    g = golden('g1_decompose')
    feats = fm.sym('g1/feats', (2, 512, 24), 1.0)
    bb, bn = fm.sym('g1/bb', (1, 7, 512), 1.0), fm.sym('g1/bn', (1, 4, 512), 1.0)
    f2d = feats.permute(0, 2, 1).reshape(48, 512).contiguous().to(DEV)
    sb = F.normalize(bb[0], p=2, dim=-1).to(DEV)
    sn = F.normalize(bn[0], p=2, dim=-1).to(DEV)
    bg = torch.empty_like(f2d)
    proj = ops.pop_decompose_into(f2d, sb.contiguous(), bg)
    got_proj = proj.view(2, 24, 7).permute(0, 2, 1).cpu().numpy()
    got_bg = bg.view(2, 24, 512).permute(0, 2, 1).cpu().numpy()
    assert np.abs(got_proj - g['proj']).max() <= 2e-6 * np.abs(g['proj']).max()
    assert np.abs(got_bg - g['bg'][:, 0]).max() <= 2e-6 * np.abs(g['bg']).max()
    # rank-1 foreground components p_k * s_k (never materialised by the kernels): rebuilt from proj for the golden's sub-sampled channels
    fg = got_proj[:, :, None, :] * sb.cpu().numpy()[None, :, ::32, None]
    assert np.abs(fg - g['fg_sub']).max() <= 2e-6 * np.abs(g['fg_sub']).max()
    bg2 = torch.empty_like(f2d)
    proj2 = ops.pop_decompose_into(f2d, torch.cat([sb, sn]).contiguous(), bg2)
    got_bg2 = bg2.view(2, 24, 512).permute(0, 2, 1).cpu().numpy()
    assert np.abs(got_bg2 - g['bg2'][:, 0]).max() <= 2e-6 * np.abs(g['bg2']).max()
    fgn = proj2.view(2, 24, 11).permute(0, 2, 1).cpu().numpy()[:, 7:, None, :] * sn.cpu().numpy()[None, :, ::32, None]
    assert np.abs(fgn - g['fgn_sub']).max() <= 2e-6 * np.abs(g['fgn_sub']).max()
    # bf16 storage mode of the same kernel: bounded by the bf16 rounding of the stored residual
    bgh = torch.empty((48, 512), dtype=torch.bfloat16, device=DEV)
    ph = ops.pop_decompose_into(f2d.to(torch.bfloat16), sb.contiguous(), bgh)
    assert np.abs(ph.view(2, 24, 7).permute(0, 2, 1).cpu().numpy() - g['proj']).max() <= 2e-2 * np.abs(g['proj']).max()


# --------------------------------------------------------------------------------------------- (d) argmax / pseudo-labels, same box
def test_argmax_and_pseudo_labels_vs_same_box_oracle(hip):
    """'argmax masks bit-exact' (north_star) measured where it can be exact: the fp32 HIP path against the CPU oracle evaluated on THIS
    machine.  0 differing pixels outside a 1e-5 (of logit scale) top-2 margin for the upsampled argmax (eval_base.py:168-170) and for the
    in-place pseudo-labels of forward_novel (pspnet_pop.py:221-231); the differing-pixel counts against the cross-machine goldens
    G6 / G7 are printed and bounded."""
    from oracle import pop_oracle as po
    from segland_amd import ops
    # ---- base model, eval: logits -> upsample(align_corners=True) -> argmax
    m = _model(dtype=torch.float32, criterion=False)
    fm.load_formula_weights(m)
    m = m.to(DEV).eval()
    o = fm.load_formula_weights(po.PopOracle(n_base=7, backbone='resnet50')).eval()
    img = fm.formula_image(2, 512, 512, 'g6/img')
    with torch.no_grad():
        lg = m(img.to(DEV))
        lo = o(img)
    am = ops.upsample_argmax(lg.contiguous(), (512, 512)).cpu().numpy()
    up = F.interpolate(lo, size=(512, 512), mode='bilinear', align_corners=True)
    ref = up.argmax(1).numpy().astype(np.uint8)
    top2 = up.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    scale = float(lo.abs().max())
    diff = am != ref
    n_clear = int((diff & (margin > 1e-5 * scale)).sum())
    print('argmax vs same-box oracle: %d / %d pixels differ, %d outside the 1e-5 margin; logits max err %.2e of scale'
          % (diff.sum(), diff.size, n_clear, float((lg.cpu() - lo).abs().max()) / scale))
    assert n_clear == 0
    # bit-exact label map against the oracle evaluated on this machine: 0 of 524 288 pixels in every run of rounds 2-4 (profiles/r4_parity_log.txt); a pixel inside the
    # 1e-5 margin could in principle flip with another host's summation order -- the message then says how many and how close
    assert diff.sum() == 0, '%d argmax pixels differ from the same-box oracle (all within the 1e-5 top-2 margin)' % diff.sum()
    # the train-mode argmax of golden G6 (cross-machine): count and bound
    g = golden('g6_full_r50')
    mt = _model(dtype=torch.float32, criterion=False)
    fm.load_formula_weights(mt)
    mt = mt.to(DEV).train()
    with torch.no_grad():
        lt = mt(img.to(DEV))
    amt = ops.upsample_argmax(lt.contiguous(), (512, 512)).cpu().numpy()
    upg = F.interpolate(torch.from_numpy(g['logits']), size=(512, 512), mode='bilinear', align_corners=True)
    t2 = upg.topk(2, dim=1).values
    mg = (t2[:, 0] - t2[:, 1]).numpy()
    dg = amt != g['argmax']
    sg = float(np.abs(g['logits']).max())
    print('argmax vs golden G6 (cross-machine, train-mode BN): %d / %d pixels differ, largest margin among them %.2e of scale'
          % (dg.sum(), dg.size, float(mg[dg].max() / sg) if dg.any() else 0.0))
    assert dg.sum() <= 600 and (not dg.any() or mg[dg].max() <= 2e-3 * sg)
    # ---- ft model: pseudo-labels written into mask_b
    kw = dict(n_base=7, is_ft=True, n_novel=4, backbone='resnet50', dilated=True, os=8)
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    mf = GFSS_Model(criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.float32, **kw)
    fm.load_formula_weights(mf)
    of = fm.load_formula_weights(po.PopOracle(criterion=po.OrthLossOracle(255), **kw))
    mf.init_cls_n(); po.init_cls_n(of)
    with torch.no_grad():
        for (k, p), (_, q) in zip(mf.classifier_n.named_parameters(), of.classifier_n.named_parameters()):
            d = fm.sym('g7/cn/' + k, tuple(p.shape), 0.01)
            p.add_(d); q.add_(d)
    mf = mf.to(DEV)
    img1, img_b = fm.formula_image(1, 512, 512, 'g7/img'), fm.formula_image(1, 512, 512, 'g7/img_b')
    mask = fm.formula_mask(1, 512, 512, 4, 'g7/mask', ignore_rows=0, lo=8); mask[mask == 8] = 255
    mask_b = fm.formula_mask(1, 512, 512, 8, 'g7/mask_b', ignore_rows=0)
    mb_gpu, mb_cpu = mask_b.clone().to(DEV), mask_b.clone()
    mf.train_mode(); po.train_mode(of)
    mf(img1.to(DEV), mask.to(DEV), img_b.to(DEV), mb_gpu)
    of(img1, mask, img_b, mb_cpu)
    # margin of the oracle's novel-head argmax at the pixels that differ
    of.criterion = None
    with torch.no_grad():
        preds_o = of(img1, mask, img_b, mask_b.clone())
    p2 = torch.cat([preds_o[1:, 0:1], preds_o[1:, 8:]], 1)
    up2 = F.interpolate(p2, size=(512, 512), mode='bilinear', align_corners=True)
    tt = up2.topk(2, dim=1).values
    mg2 = (tt[:, 0] - tt[:, 1]).numpy()
    dd = (mb_gpu.cpu() != mb_cpu).numpy()
    s2 = float(p2.abs().max())
    print('pseudo-labels vs same-box oracle: %d pixels differ, %d outside the 1e-5 margin'
          % (dd.sum(), int((dd & (mg2 > 1e-5 * s2)).sum())))
    assert int((dd & (mg2 > 1e-5 * s2)).sum()) == 0
    assert dd.sum() == 0, '%d pseudo-label pixels differ from the same-box oracle (all within the 1e-5 top-2 margin)' % dd.sum()
    g7 = golden('g7_ft')
    n7 = int((mb_gpu.cpu().numpy().astype(np.uint8) != g7['mask_b_new']).sum())
    print('pseudo-labels vs golden G7 (cross-machine): %d pixels differ' % n7)
    assert n7 <= 8


# --------------------------------------------------------------------------------------------- (b) C2 train-mode gate in bf16
def _group_cosines(model, oracle):
    from segland_amd.utils.pyt_utils import get_parameters
    mine, ref = dict(model.named_parameters()), dict(oracle.named_parameters())
    groups = {'backbone': [], 'head_bias': [], 'head_other': []}
    for k, p in mine.items():
        if p.grad is None:
            continue
        grp = 'backbone' if 'backbone' in k else ('head_bias' if 'bias' in k else 'head_other')    # utils/pyt_utils.py:216-249
        groups[grp].append(k)
    out = {}
    for grp, keys in groups.items():
        a = torch.cat([mine[k].grad.detach().float().cpu().reshape(-1) for k in keys]).double()
        b = torch.cat([ref[k].grad.detach().reshape(-1) for k in keys]).double()
        out[grp] = (float((a @ b) / (a.norm() * b.norm())), float((a - b).norm() / b.norm()))
    return out


def _structured_batch(B, size, seed):
    """Tiles whose colour statistics depend on the label (block-structured masks, class-dependent mean colour + noise): the loss gradient is
    a coherent sum over pixels, as on real land-cover tiles, not the sqrt(N) residual of random labels."""
    mask = fm.formula_mask(B, size, size, 8, 'c2/mask%d' % seed, block=32, ignore_rows=size // 10)
    g = torch.Generator().manual_seed(seed)
    color = torch.randn(8, 3, generator=g)
    idx = mask.clone(); idx[idx == 255] = 0
    img = color[idx].permute(0, 3, 1, 2).contiguous() + 0.5 * torch.randn(B, 3, size, size, generator=g)
    return img, mask


@pytest.mark.timeout(900)
@pytest.mark.parametrize('size', [512])            # the bench shape itself (round 3 also ran 256 x 256: the same gates on a quarter of the pixels, 30 s of the suite)
def test_c2_train_mode_bf16_gate(hip, size):
    """Config C2 (the bench configuration: R50, bf16, batch 16, TRAIN-mode BatchNorm) against the fp32 CPU oracle on this machine.

    Why not at He-initialisation: tools/exp_bf16_conditioning.py (result in profiles/r2_bf16_conditioning.txt) shows with the ORACLE ALONE that
    ONE bf16 rounding of the stem output of an untrained ResNet-50 decorrelates the backbone gradient (cosine 0.26-0.42 against the
    unrounded run): the gradient of an untrained network on any data is the incoherent residual of a sum over pixels, and every ReLU-mask
    flip moves whole terms.  No bf16 implementation can pass a cosine gate there; the He-init numbers are printed below for the record.
    After 16 optimisation steps on a structured batch the gradient is a coherent sum and the same experiment gives cosine 0.9989
    for an oracle that rounds EVERY conv / BN output and gradient to bf16 -- that is where the gate is set:
      * weights: He-uniform init (torch default, fixed seed), 16 train_base.py iterations (AdamW, lr 1e-3) in the exact-fp32 HIP mode,
        conv weights then rounded to bf16 on both sides;
      * batch 16 tiles of 256x256 (16 x 32 x 32 samples per channel in the deepest BN; the oracle needs seconds) and -- the bench shape itself --
        of 512x512 (the CPU oracle then holds ~28 GB of fp32 activations and needs about a minute per forward + backward: run when the host has the
        memory, He-init record skipped);
      * gates: step loss within 1e-2 relative, gradient cosine >= 0.99 for each of the three optimizer parameter groups
        (utils/pyt_utils.py:216-249); the exact-fp32 HIP mode must reach >= 0.9995 on the same state."""
    from oracle import pop_oracle as po
    from segland_amd.optim import AdamW
    from segland_amd.train_base import train_iteration
    from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    if size == 512:
        import psutil
        avail = psutil.virtual_memory().available / 2 ** 30
        if avail < 56:
            pytest.skip('the fp32 CPU oracle at batch 16, 512x512 needs ~28 GB of host memory (+ margin); %.0f GB available' % avail)
    img, mask = _structured_batch(16, size, seed=5)
    gi, gm = img.to(DEV), mask.to(DEV)

    def compare(state, tag, gate):
        o = po.PopOracle(n_base=7, criterion=po.OrthLossOracle(255), backbone='resnet50')
        o.load_state_dict(state, strict=True)
        o.train()
        do = o(img, mask)
        do['total_loss'].backward()
        res = {}
        for dt in (torch.float32, torch.bfloat16):
            m = _model(dtype=dt)
            m.load_state_dict(state, strict=True)
            m = m.to(DEV).train()
            from segland_amd import ops
            watch = gate and dt == torch.bfloat16 and size == 512
            if watch:
                ops.PROFILER.start()
            d = m(gi, gm)
            d['total_loss'].backward()
            if watch:
                # round 6: the gated bf16 run IS the bench's default dispatch -- every conv launch of this forward + backward attributed to the
                # kernel family it ran on (the same query bench.py uses), and the BatchNorm-backward fusions taken (4 stand-alone reduce launches of 58 BatchNorms)
                fams, tb = ops.PROFILER.stop(), ops.PROFILER.stop_bytes()
                need = {'conv_gemm_p9_kernel<bf16, 256, 256>', 'conv_gemm_p8_kernel<bf16, 256, 256>', 'conv_gemm_sk_kernel<bf16, 256, 64>', 'conv_wgrad3_kernel',
                        'conv_wgrad_glds_kernel<bf16, 256, 256>', 'conv_wgrad_glds_kernel<bf16, 128, 256>', 'conv_wgrad_c64k3_kernel', 'conv_wgrad_c64p_kernel',
                        'conv_c64k3_kernel<bf16, 16, 16>'}
                print('C2 bf16 run: conv launches by family: %s; bn_bwd_reduce launches %d' % (', '.join('%s x%d' % (k, v['calls']) for k, v in sorted(fams.items())),
                                                                                           tb.get('bn_bwd_reduce', {}).get('calls', 0)))
                assert need <= set(fams), 'the gated bf16 run did not take the bench kernels: missing %s' % sorted(need - set(fams))
                assert fams['conv_wgrad3_kernel']['calls'] == 13 and fams['conv_gemm_p9_kernel<bf16, 256, 256>']['calls'] == 20, {k: v['calls'] for k, v in fams.items()}
                assert tb.get('bn_bwd_reduce', {}).get('calls', 0) <= 4, tb.get('bn_bwd_reduce')
            rel = abs(float(d['seg_loss'].detach()) - float(do['seg_loss'].detach())) / abs(float(do['seg_loss'].detach()))
            cos = _group_cosines(m, o)
            print('C2 %dx%d %s %-8s seg_loss hip %.5f oracle %.5f (rel %.1e) | gradient cosine / rel.L2: %s' % (
                size, size, tag, str(dt)[6:], float(d['seg_loss'].detach()), float(do['seg_loss'].detach()), rel,
                ', '.join('%s %.5f / %.3f' % (k, c, l) for k, (c, l) in cos.items())))
            res[dt] = (rel, cos)
        if gate:
            rel, cos = res[torch.float32]
            assert rel <= 1e-4 and all(c >= 0.9995 for c, _ in cos.values()), ('fp32 mode', rel, cos)
            rel, cos = res[torch.bfloat16]
            assert rel <= 1e-2, rel
            assert all(c >= 0.99 for c, _ in cos.values()), ('bf16 mode', cos)

    torch.manual_seed(1234)
    m = _model(dtype=torch.float32)
    _round_weights_to_bf16_(m)
    if size == 256:
        compare({k: v.clone() for k, v in m.state_dict().items()}, 'He-init (not gated)', gate=False)
    m = m.to(DEV).train()
    opt = AdamW(get_parameters(m, lr=1e-3), lr=1e-3, weight_decay=1e-4)
    scaler = NativeScalerWithGradNormCount()
    for _ in range(16):
        d, _ = train_iteration(m, opt, scaler, gi, gm, double_step=False)
    print('C2 after 16 fp32-mode steps: seg_loss %.4f' % float(d['seg_loss'].detach()))
    _round_weights_to_bf16_(m)
    compare({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, 'after 16 steps', gate=True)


# --------------------------------------------------------------------------------------------- (a) ResNet-101 bf16 at the bench size
def _hip_group_cosines(m_a, m_b):
    out = {}
    for grp, sel in (('backbone', lambda k: 'backbone' in k), ('head_bias', lambda k: 'backbone' not in k and 'bias' in k),
                     ('head_other', lambda k: 'backbone' not in k and 'bias' not in k)):
        ka = [k for k, p in m_a.named_parameters() if sel(k) and p.grad is not None]
        a = torch.cat([dict(m_a.named_parameters())[k].grad.detach().float().reshape(-1) for k in ka]).double()
        b = torch.cat([dict(m_b.named_parameters())[k].grad.detach().float().reshape(-1) for k in ka]).double()
        out[grp] = float((a @ b) / (a.norm() * b.norm()))
    return out


@pytest.mark.timeout(900)
def test_r101_bf16_b16_512_step_and_eval(hip):
    """Config C3 per GPU (R101, bf16, batch 16, 512x512, train-mode BN).  The untrained network is ill-conditioned against ANY bf16
    rounding (see test_c2_train_mode_bf16_gate), so the state is taken after 32 train_base.py iterations in the exact-fp32 HIP mode on a
    structured batch; on that state one bf16 iteration is compared with the same iteration in fp32 mode: losses <= 2 % (measured 1e-4),
    every gradient finite, gradient cosine per optimizer parameter group >= 0.99
    (tools/exp_r101_cos.py: backbone cosine 0.90 / 0.985 / 0.997 / 0.9997 after 8 / 16 / 24 / 32 steps).  Then eval logits at batch 2, 512x512 against the CPU
    oracle on this machine: <= 5 % of the logit scale, >= 97 % argmax agreement."""
    from oracle import pop_oracle as po
    from segland_amd.optim import AdamW
    from segland_amd.train_base import train_iteration
    from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
    img, mask = _structured_batch(16, 512, seed=7)
    img, mask = img.to(DEV), mask.to(DEV)
    torch.manual_seed(99)
    m32 = _model('resnet101', dtype=torch.float32)
    _round_weights_to_bf16_(m32)
    m32 = m32.to(DEV).train()
    opt = AdamW(get_parameters(m32, lr=1e-3), lr=1e-3, weight_decay=1e-4)
    scaler = NativeScalerWithGradNormCount()
    for _ in range(32):
        d, _ = train_iteration(m32, opt, scaler, img, mask, double_step=False)
    print('R101 after 32 fp32-mode steps: seg_loss %.4f' % float(d['seg_loss'].detach()))
    _round_weights_to_bf16_(m32)
    state = {k: v.detach().clone() for k, v in m32.state_dict().items()}
    m16 = _model('resnet101', dtype=torch.bfloat16)
    m16.load_state_dict(state)
    m16 = m16.to(DEV).train()
    res = {}
    for tag, m in (('fp32', m32), ('bf16', m16)):
        m.load_state_dict(state)
        m.zero_grad(set_to_none=True)
        sgd = torch.optim.SGD(get_parameters(m, lr=0.0), lr=0.0)        # lr 0: the loop body runs, the weights stay
        d, gn = train_iteration(m, sgd, NativeScalerWithGradNormCount(), img, mask, double_step=False)
        grads = [p.grad for p in m.parameters() if p.requires_grad]
        assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads), 'non-finite or missing gradient (%s)' % tag
        res[tag] = ({k: float(v.detach()) for k, v in d.items()}, float(gn))
    for k in ('total_loss', 'seg_loss'):
        rel = abs(res['bf16'][0][k] - res['fp32'][0][k]) / abs(res['fp32'][0][k])
        print('R101 B=16 512^2 train step: %s bf16 %.5f fp32-mode %.5f rel %.2e' % (k, res['bf16'][0][k], res['fp32'][0][k], rel))
        assert rel <= 2e-2
    cos = _hip_group_cosines(m16, m32)
    print('R101 grad norm bf16 %.4f fp32-mode %.4f; gradient cosine bf16 vs fp32 mode: %s' % (res['bf16'][1], res['fp32'][1], cos))
    assert abs(res['bf16'][1] - res['fp32'][1]) <= 0.1 * res['fp32'][1]
    assert all(c >= 0.99 for c in cos.values()), cos
    # eval logits, batch 2, vs the oracle here
    o = po.PopOracle(n_base=7, backbone='resnet101')
    o.load_state_dict({k: v.detach().float().cpu() for k, v in m16.state_dict().items()}, strict=True)
    o.eval(); m16.eval()
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        lg = m16(img[:2]).float().cpu()
        lo = o(img[:2].cpu())
    err = float((lg - lo).abs().max() / lo.abs().max())
    agree = float((lg.argmax(1) == lo.argmax(1)).float().mean())
    print('R101 bf16 eval logits vs same-box oracle: max err %.3f of scale, argmax agreement %.4f' % (err, agree))
    assert err <= 5e-2 and agree >= 0.97


# --------------------------------------------------------------------------------------------- (c) the product under RCCL DDP
def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize('sync', ['force', 'inplace'])       # ('0', stock DistributedDataParallel with its copy + scale hook, ran here until round 3: the same reducer as 'inplace' minus the build's part)
def test_hip_model_under_rccl_ddp(hip, sync):
    """engine.py:71 / train_base.py:175-178 on the product: a fresh child process creates a world_size-1 RCCL group, wraps the HIP model
    with Engine.data_parallel (DDP bucket hooks x once_differentiable block Functions, gradient_as_bucket_view x the AdamW pointer table,
    the per-step weight-copy refresh) and runs two train_base.py iterations; parameters, buffers and eval logits must equal the unwrapped
    run.  sync='force': the same through nn.SyncBatchNorm with SEGLAND_SYNC_BN semantics (world 1: the all-reduce is the identity).
    sync='inplace': Engine.data_parallel(sum_gradients=True) -- sum-only all-reduce hook, block backwards writing parameter gradients straight
    into DDP's bucket views (functional.grad_dst), the 1 / world_size inside the AdamW kernel: three iterations, identical parameters, and
    the third backward must have written >= 170 of the 180 gradients in place."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='16')      # two or three children next to pytest: 256 OpenMP threads each oversubscribe the host
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ddp_child.py'), sync, str(_free_port())], env=env, capture_output=True,
                       text=True, timeout=540)
    line = [l for l in r.stdout.splitlines() if l.startswith('DDP_CHILD ')]
    assert r.returncode == 0 and line, 'child failed (rc %d):\n%s\n%s' % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    out = json.loads(line[-1][len('DDP_CHILD '):])
    print(out)
    if sync != 'inplace':                      # in-place mode: the gradients are fresh aliases of the views (checked through grads_alias_cached_views)
        assert out['bucket_view_grads'] > 100, 'gradients are not DDP bucket views'
    assert out['bn1_tracked'] == (3 if sync == 'inplace' else 2)
    tol = 2e-4 if sync == 'force' else 1e-6
    if sync == 'inplace':
        assert out['inplace_writes_last_step'] >= 170 and out['grads_alias_cached_views'] >= 170, out
    assert out['worst_param_rel'] <= tol, out
    assert out['logits_rel'] <= (1e-3 if sync == 'force' else 1e-6), out
    for (a, ga), (b, gb) in zip(out['losses'], out['ref_losses']):
        assert abs(a - b) <= 1e-5 * abs(b) + (1e-4 if sync == 'force' else 0) and abs(ga - gb) <= 1e-3 * gb


@pytest.mark.timeout(900)
def test_two_ranks_equal_one_full_batch(hip, tmp_path):
    """a 2-rank data-parallel step on a split batch == the 1-process step on the full batch.  Two processes share the one
    GPU of the box (gloo group: RCCL refuses two ranks per device); the product path is otherwise complete -- bucket_step.BucketedReplica (flat buckets,
    gradients written in place, one SUM all-reduce per bucket; issued kernel by kernel because of SyncBatchNorm), nn.SyncBatchNorm with global statistics (fp64 all-reduce of the partial sums, PPM level 1
    with ONE value per channel per rank included), AdamW with 1 / world_size in its kernel.  One backward: loss, all gradients and the running
    statistics agree to reduction-order tolerance; three train_base.py iterations on top stay together."""
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='16')      # two or three children next to pytest: 256 OpenMP threads each oversubscribe the host
    child = os.path.join(ROOT, 'tests', 'ddp2_child.py')
    outs = [str(tmp_path / 'two.pt'), str(tmp_path / 'one.pt')]
    procs = [subprocess.Popen([sys.executable, child, str(r), port, outs[0]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    logs = [p.communicate(timeout=800)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), 'rank failed:\n' + '\n----\n'.join(l[-3000:] for l in logs)
    r = subprocess.run([sys.executable, child, '-1', port, outs[1]], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    two, one = torch.load(outs[0]), torch.load(outs[1])
    # (1) one forward/backward: loss, every gradient and the updated running statistics, exact up to the order of the reductions
    assert abs(two['loss0'] - one['loss0']) <= 1e-4 * abs(one['loss0']), (two['loss0'], one['loss0'])        # measured: equal to the last digit
    assert two['grads0'].keys() == one['grads0'].keys() and len(one['grads0']) >= 170
    num = sum(float(((two['grads0'][k] - g) ** 2).sum()) for k, g in one['grads0'].items())
    den = sum(float((g ** 2).sum()) for g in one['grads0'].values())
    worst_g, key_g = max((float((two['grads0'][k] - g).norm() / max(float(g.norm()), 1e-20)), k) for k, g in one['grads0'].items())
    print('gradients after one backward: global relative L2 %.3e, worst tensor %.3e (%s)' % ((num / den) ** 0.5, worst_g, key_g))
    # the train-mode BN chain of the formula-weight network amplifies a 1e-7 difference of the batch statistics ~1e3..1e4 x:
    # measured 1e-3; a missing 1 / world_size, a per-rank count or a dropped rank would be O(1)
    assert (num / den) ** 0.5 <= 5e-3 and worst_g <= 2e-2, (worst_g, key_g)
    for k, b in one['stats0'].items():
        assert float((two['stats0'][k] - b).abs().max()) <= 1e-3 * max(float(b.abs().max()), 1e-3), k
    # (1b) two iterations with a plain SGD update through the same step machinery (bucket all-reduces, clip coefficient with the 1 / world_size, double step): the
    # summed update is linear in the gradients, so it is BOUNDED -- a factor that slips in at the second step (1 / world_size, a stale or doubly reduced bucket)
    # would show as tens of per cent, not as the ~1e-3 of the reduction order
    assert two['sgd_update'].keys() == one['sgd_update'].keys() and len(one['sgd_update']) >= 170
    un = sum(float(((two['sgd_update'][k] - u) ** 2).sum()) for k, u in one['sgd_update'].items())
    ud = sum(float((u ** 2).sum()) for u in one['sgd_update'].values())
    print('summed update of two SGD iterations: global relative L2 %.3e; losses 2 ranks %s, 1 process %s' % ((un / ud) ** 0.5, two['sgd_losses'], one['sgd_losses']))
    assert ud > 0 and (un / ud) ** 0.5 <= 0.1          # measured 3.2e-2 (the BN chain amplifies the 7e-4 of the first step); a factor 2 or 1 / 2 at the second step would be >= 0.25
    for (a, ga), (b, gb) in zip(two['sgd_losses'], one['sgd_losses']):
        assert abs(a - b) <= 3e-3 * abs(b) and abs(ga - gb) <= 5e-2 * gb
    # (2) three train_base.py iterations on top (Adam's sign-like first steps amplify the last bits: sanity bounds only)
    print('losses (total, grad norm)  2 ranks:', two['losses'], ' 1 process:', one['losses'])
    for (a, ga), (b, gb) in zip(two['losses'], one['losses']):
        assert abs(a - b) <= 3e-3 * abs(b) and abs(ga - gb) <= 0.1 * gb
    worst, key = 0.0, ''
    for k, b in one['sd'].items():
        if k.endswith('num_batches_tracked'):
            assert torch.equal(two['sd'][k], b)
            continue
        e = float((two['sd'][k] - b).abs().max() / max(float(b.abs().max()), 1e-12))
        if e > worst:
            worst, key = e, k
    print('after three iterations: worst parameter / buffer difference %.3e (%s)' % (worst, key))
    # (no bound on `worst`: lr * sign(g) steps on weights of magnitude ~lr make it O(1) for some tensors; the losses and logits bound the drift)
    assert float((two['logits'] - one['logits']).abs().max() / one['logits'].abs().max()) <= 5e-2


# --------------------------------------------------------------------------------------------- ADVICE regressions
def test_eval_coefficients_follow_train_mode_statistics(hip):
    """ADVICE r1 (functional.py:182): BN with FROZEN affine parameters run in train mode (get_parameters(fix_bn=True), BN recalibration)
    updates its running statistics through raw pointers; the cached eval scale/shift -- and the HIP graph of the frozen feature
    extractor built on them -- must follow.  eval -> train-mode forward -> eval, compared with the oracle doing the same."""
    from oracle import pop_oracle as po
    m = _model(dtype=torch.float32, criterion=False)
    fm.load_formula_weights(m)
    o = fm.load_formula_weights(po.PopOracle(n_base=7, backbone='resnet50'))
    for p in m.parameters():
        p.requires_grad = False
    m = m.to(DEV)
    img = fm.formula_image(2, 128, 128, 'recal/img')
    img2 = fm.formula_image(2, 128, 128, 'recal/img2')
    m.eval(); o.eval()
    with torch.no_grad():
        a0, b0 = m(img.to(DEV)).cpu(), o(img)                     # fills the eval-coefficient cache and captures the feature graph
    assert float((a0 - b0).abs().max()) <= 2e-3 * float(b0.abs().max())
    m.train(); o.train()
    with torch.no_grad():
        m(img2.to(DEV)); o(img2)                                   # recalibration forward: running statistics move, no optimizer step
    m.eval(); o.eval()
    with torch.no_grad():
        a1, b1 = m(img.to(DEV)).cpu(), o(img)
    assert float((b1 - b0).abs().max()) > 1e-2 * float(b0.abs().max()), 'the recalibration forward did not change the oracle output'
    assert float((a1 - b1).abs().max()) <= 2e-3 * float(b1.abs().max()), 'eval after a train-mode forward used stale BN coefficients'


def test_ft_pop_with_loader_workers_and_pinned_memory(hip, tmp_path):
    """ADVICE r1 (pspnet_pop.py:132): the HIP-graph capture of the frozen feature extractor runs while the DataLoader's worker processes
    and pin-memory thread are alive (--num-workers 2, pin_memory=True: the drivers' defaults)."""
    import glob
    from segland_amd import ft_pop
    snap = str(tmp_path / 'snap_ft_workers')
    ft_pop.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic', '--batch-size', '1', '--input-size', '128,128',
                 '--base-size', '128,128', '--num-epoch', '1', '--learning-rate', '1e-3', '--print-frequency', '5', '--snapshot-dir', snap,
                 '--num-workers', '2', '--restore-from', '/nonexistent', '--allow-random-init', '--random-seed', '123', '--freeze-backbone', '--fix-bn'])
    assert glob.glob(os.path.join(snap, 'epoch_0_123.pth'))


def test_missing_checkpoint_raises(hip, tmp_path):
    from segland_amd import eval_base
    with pytest.raises(FileNotFoundError):
        eval_base.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic', '--base-size', '128,128',
                        '--restore-from', str(tmp_path / 'missing.pth')])


def test_adamw_mixed_step_counts(hip):
    """ADVICE r1 (optim.py:48): torch.optim.AdamW keeps a step count per parameter; a parameter that first receives a gradient on a later
    iteration must not abort the step.  amsgrad / maximize raise instead of being ignored."""
    from segland_amd.optim import AdamW
    shapes = [(33, 7), (513,), (64, 16, 3, 3)]
    mine = [fm.sym('adam2/p%d' % i, s, 1.0).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    o1, o2 = AdamW(mine, lr=1e-2, weight_decay=1e-2), torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2, foreach=True)
    for it in range(4):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i == 1 and it < 2:
                a.grad = b.grad = None                     # this one joins at iteration 2
                continue
            g = fm.sym('adam2/g%d_%d' % (it, i), tuple(a.shape), 0.5).to(DEV)
            a.grad, b.grad = g.clone(), g.clone()
        o1.step(); o2.step()
    for a, b in zip(mine, ref):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    assert float(o1.state_dict()['state'][1]['step']) == 2 and float(o1.state_dict()['state'][0]['step']) == 4
    with pytest.raises(NotImplementedError):
        AdamW(mine, amsgrad=True)


def test_per_gpu_batch_one_needs_sync_bn(hip, monkeypatch):
    """ADVICE r1 (functional.py:123): per-GPU batch 1 puts ONE value per channel into the PPM's 1x1 pyramid level.  The default per-GPU
    BatchNorm raises like F.batch_norm does; with SEGLAND_SYNC_BN the GLOBAL count decides, as in torch's SyncBatchNorm, and the
    step runs (fake 2-rank all-reduce: two ranks holding the same tile)."""
    import torch.distributed as dist
    from segland_amd import functional as sf
    img, mask = _batch(1, 128, seed=3)
    m = _model(dtype=torch.float32, norm_layer=nn.SyncBatchNorm).to(DEV).train()
    with pytest.raises(ValueError, match='more than 1 value per channel'):
        m(img.to(DEV), mask.to(DEV))
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda *a, **k: 2)
    monkeypatch.setattr(dist, 'all_reduce', lambda t, *a, **k: t.mul_(2))
    sf.set_sync_bn('1')
    try:
        d = m(img.to(DEV), mask.to(DEV))
        d['total_loss'].backward()
    finally:
        sf.set_sync_bn('0')
    assert bool(torch.isfinite(d['total_loss'])) and bool(torch.isfinite(m.base_emb.grad).all())


# --------------------------------------------------------------------------------------------- f-2: OEM tile preparation on the GPU
def test_g17_tile_preparation_gpu(hip):
    """csrc/augment.hip against golden G17 (dataset/base_dataset.py:29-175 + oem.py:113-133 executed by the reference): bit-exact float
    images and integer labels for tiles smaller than / equal to / larger than the crop, every flip / rot90 combination, a batch in one launch."""
    from oracle import data_oracle as do
    from segland_amd.dataset.augment import TileAugmenter, remap_lut
    g = golden('g17_oem_tiles')
    aug = TileAugmenter((64, 64), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 255, device=DEV)
    tiles, params, want = [], [], []
    for tag, (H, W) in (('small', (50, 70)), ('exact', (64, 64)), ('large', (100, 90))):
        img = (fm.uniform01('g17/%s/img' % tag, H * W * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(H, W, 3).numpy()
        lab = (fm.uniform01('g17/%s/lab' % tag, H * W) * 12).floor().to(torch.uint8).reshape(H, W).numpy()
        lab[:7] = 255
        for rep in range(3):
            p = g['%s_%d_prm' % (tag, rep)].tolist()
            tiles.append((img, lab)); params.append((p[0], p[1], bool(p[2]), p[3])); want.append((tag, rep))
    out, lab_out = aug.prepare(tiles, params)
    for b, (tag, rep) in enumerate(want):
        assert np.array_equal(out[b].cpu().numpy()[:, ::3, ::3], g['%s_%d_img' % (tag, rep)]), (tag, rep)
        assert np.array_equal(lab_out[b].cpu().numpy().astype(np.uint8), g['%s_%d_lbl' % (tag, rep)]), (tag, rep)
    # every flip / rotation against the oracle, with a re-indexing table and a non-trivial mean / std
    lut = remap_lut(set(range(1, 8)), set(range(8, 12)), True, True)
    aug2 = TileAugmenter((48, 48), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), 255, lut=lut, device=DEV)
    img = (fm.uniform01('g17x/img', 60 * 52 * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(60, 52, 3).numpy()
    lab = (fm.uniform01('g17x/lab', 60 * 52) * 12).floor().to(torch.uint8).reshape(60, 52).numpy()
    prm = [(5, 3, f, k) for f in (False, True) for k in range(4)] + [(20, 10, True, 1)]
    o2, l2 = aug2.prepare([(img, lab)] * len(prm), prm)
    for b, p in enumerate(prm):
        io, lo = do.prepare_tile(img, lab, (48, 48), *p, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
        assert np.array_equal(o2[b].cpu().numpy(), io), p
        assert np.array_equal(l2[b].cpu().numpy(), lut[lo.astype(np.uint8)].astype(np.int64)), p
    # normalize() with the BaseDataset default statistics on a whole (uncropped) tile, as stored by the reference
    img = (fm.uniform01('g17/norm/img', 40 * 48 * 3) * 256).floor().clamp(0, 255).to(torch.uint8).reshape(40, 48, 3).numpy()
    aug3 = TileAugmenter((40, 48), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), 255, device=DEV)
    o3, l3 = aug3.prepare([(img, None)], [(0, 0, False, 0)])
    assert l3 is None and np.array_equal(o3[0].cpu().numpy(), g['norm_img'])


def test_train_base_on_raw_tiles_with_workers(hip, tmp_path):
    """Row f-2 through the driver: DataLoader workers DECODE TIFF files (`synthetic_tiff`: the real dataset/oem.py readers on a generated OpenEarthMap-shaped
    directory, dataset/tiff.py) and hand over the raw uint8 tiles as shared-memory tensors + the reference's random draws, one augment launch per batch on the GPU,
    training and the end-of-run validation on whole tiles."""
    import glob
    from segland_amd import train_base
    snap = str(tmp_path / 'snap_raw')
    train_base.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic_tiff', '--batch-size', '4', '--input-size', '128,128',
                     '--base-size', '160,160', '--num-epoch', '36', '--start-epoch', '35', '--learning-rate', '1e-4', '--print-frequency', '8', '--snapshot-dir', snap,
                     '--num-workers', '2', '--restore-from', '/nonexistent', '--allow-random-init', '--fp16'])
    assert glob.glob(os.path.join(snap, 'epoch_36.pth')) and glob.glob(os.path.join(snap, 'best.pth'))


def test_g19_pair_reader_gpu(hip, tmp_path):
    """Row f-2: the product's fine-tune pair path end to end -- PairReader (lists, draws) -> pair_collate -> PairAugmenter (ONE launch for the 2B
    tiles) -- bit-exact against golden G19, i.e. against the tensors dataset/oem_ft.py of the reference returned for the same seeds."""
    import random
    from g19_common import product_reader
    from segland_amd.dataset.oem_ft import PairAugmenter, pair_collate
    g = golden('g19_oem_ft')
    novel_ids = g['novel_ids'].tolist()
    for filt in (False, True):
        tag = 'f%d' % int(filt)
        random.seed(7); np.random.seed(7)
        ds = product_reader(tmp_path / tag, filt, novel_ids=novel_ids)()
        random.seed(21); np.random.seed(21)
        samples = [ds[i] for i in (0, 5, len(ds) - 1)]
        ds.update_base_list()
        samples += [ds[i] for i in (1, 2)]
        pairs, params, ids = pair_collate(samples)
        aug = PairAugmenter((64, 64), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 255, device=DEV)         # the statistics G19's pairs were stored with
        img, mask, img_b, mask_b = aug.prepare(pairs, params)
        assert mask.dtype == torch.int64 and img.shape == (5, 3, 64, 64)
        for k in range(5):
            assert ids[k] == g['%s_p%d_id' % (tag, k)][0]
            assert np.array_equal(img[k].cpu().numpy()[:, ::4, ::4], g['%s_p%d_img' % (tag, k)]), (tag, k)
            assert np.array_equal(mask[k].cpu().numpy().astype(np.uint8), g['%s_p%d_lbl' % (tag, k)]), (tag, k)
            assert np.array_equal(img_b[k].cpu().numpy()[:, ::4, ::4], g['%s_p%d_imgb' % (tag, k)]), (tag, k)
            assert np.array_equal(mask_b[k].cpu().numpy().astype(np.uint8), g['%s_p%d_lblb' % (tag, k)]), (tag, k)
    # the reader's own default statistics (BaseDataset's ImageNet mean / std: oem_ft.py never overrides them)
    random.seed(5); np.random.seed(5)
    ds = product_reader(tmp_path / 'dflt', False, novel_ids=novel_ids)()
    assert ds.base_id_list == g['default_base'].tolist()
    s = ds[3]
    img, _, img_b, _ = ds.augmenter(DEV).prepare(*pair_collate([s])[:2])
    assert np.array_equal(img[0].cpu().numpy()[:, ::4, ::4], g['default_img']) and np.array_equal(img_b[0].cpu().numpy()[:, ::4, ::4], g['default_imgb'])


def test_ft_pop_on_raw_pairs_with_workers(hip, tmp_path):
    """ft_pop end to end on the raw pair format read from TIFF files (`--dataset synthetic_tiff`: the real dataset/oem_ft.py pair reader for training -- class lists
    built by decoding every label file -- and dataset/oem.py tiles for the validation): DataLoader workers decode + draw, PairCollate, one augment launch per batch, the
    graphed fine-tune step, validation."""
    import glob
    from segland_amd import ft_pop, graph_step
    before = dict(graph_step.STATS)
    snap = str(tmp_path / 'snap_ft_raw')
    ft_pop.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic_tiff', '--batch-size', '2', '--input-size', '128,128',
                 '--base-size', '128,128', '--num-epoch', '2', '--learning-rate', '1e-3', '--print-frequency', '4', '--snapshot-dir', snap, '--shot', '2',
                 '--num-workers', '2', '--restore-from', '/nonexistent', '--allow-random-init', '--random-seed', '123', '--freeze-backbone', '--update-base'])
    d = {k: graph_step.STATS[k] - before[k] for k in before}
    print(d)
    assert d['failures'] == 0 and d['replays'] >= 4
    assert glob.glob(os.path.join(snap, 'epoch_1_123.pth'))


def test_eval_base_on_raw_tiles_labeled_and_unlabeled(hip, tmp_path, monkeypatch):
    """eval_base on the raw-tile datasets (Engine installs raw_collate: a batch is lists of numpy tiles): labeled tiles are scored, unlabeled test
    tiles (label None -- when the reference writes its .mat files, eval_base.py:178-191) are only predicted and dumped."""
    import scipy.io
    from segland_amd import eval_base
    from segland_amd.dataset import synthetic_raw
    args = ['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic_raw', '--base-size', '128,160', '--test-batch-size', '2', '--fp16',
            '--restore-from', '/nonexistent', '--allow-random-init', '--num-workers', '2']
    res = eval_base.main(args + ['--save-path', str(tmp_path / 'lab')])
    assert 123 in res and np.isfinite(res[123][2])
    orig = synthetic_raw.GFSSegVal.__getitem__

    def unlabeled(self, i):
        img, _, prm, id_ = orig(self, i)
        return img, None, prm, id_
    monkeypatch.setattr(synthetic_raw.GFSSegVal, '__getitem__', unlabeled)
    out = tmp_path / 'unl'
    eval_base.main(args[:-2] + ['--num-workers', '0', '--save-path', str(out), '--save-prob'])
    mats = sorted(os.listdir(out / 'prob_123'))
    assert len(mats) == 8 and scipy.io.loadmat(str(out / 'prob_123' / mats[0]))['outputs'].shape == (1, 8, 128, 160)      # base evaluation: bg + 7 base classes


# --------------------------------------------------------------------------------------------- f-3: probability dumps and their fusion
def test_g18_fusion_and_probability_dump(hip, tmp_path):
    """sl_fuse_argmax bit-exact against golden G18 (the reference's fusemat.py run on three models' dumps, ties included); sl_upsample_logits
    against F.interpolate(align_corners=True) (eval_base.py:168); then the whole tool chain: eval_base --save-prob twice -> segland_amd.fusemat."""
    import scipy.io
    from PIL import Image
    from segland_amd import eval_base, fusemat, ops
    g = golden('g18_fusion')
    for tile in ('a', 'b'):
        maps = []
        for m in range(3):
            arr = fm.sym('g18/m%d/%s' % (m, tile), (1, 8, 64, 64), 3.0)
            if tile == 'b':
                arr[:, :, :8] = torch.round(arr[:, :, :8])
            maps.append(arr[0].contiguous().to(DEV))
        assert np.array_equal(ops.fuse_argmax(maps).cpu().numpy(), g['fused_' + tile]), tile
    lg = fm.sym('up/logits', (2, 12, 16, 20), 2.0)
    up = ops.upsample_logits(lg.to(DEV), (100, 130)).cpu()
    ref = F.interpolate(lg, size=(100, 130), mode='bilinear', align_corners=True)
    assert float((up - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    from segland_amd.networks.pspnet_pop import GFSS_Model
    outs = []
    for run in range(2):
        torch.manual_seed(run)
        m = GFSS_Model(n_base=7, backbone='resnet50', dilated=True, os=8, pretrained_model=None)
        ck = str(tmp_path / ('base%d.pth' % run))
        torch.save({'module.' + k: v for k, v in m.state_dict().items()}, ck, _use_new_zipfile_serialization=False)
        out = str(tmp_path / ('out%d' % run))
        eval_base.main(['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic', '--base-size', '128,128', '--restore-from', ck,
                        '--save-path', out, '--random-seed', '123', '--save-prob'])
        outs.append(os.path.join(out, 'prob_123'))
    mats = [scipy.io.loadmat(os.path.join(outs[r], '0.mat'))['outputs'] for r in range(2)]
    assert mats[0].shape[0] == 1 and mats[0].shape[2:] == (128, 128) and not np.array_equal(mats[0], mats[1])
    res = fusemat.fuse(outs, str(tmp_path / 'fused'), size=(256, 256))
    from oracle import data_oracle as do
    assert np.array_equal(res['0.mat'], do.fuse_probability_maps([mats[0][0], mats[1][0]]))
    png = np.array(Image.open(str(tmp_path / 'fused' / '0.png')))
    assert png.shape == (256, 256) and np.array_equal(png[::2, ::2], res['0.mat'])


# --------------------------------------------------------------------------------------------- f-4: true resume
def test_continue_resumes_training_state(hip, tmp_path):
    """`-c/--continue FILE` (parsed and ignored by the reference, engine.py:62-65): two epochs in one run == one epoch, save, continue for the
    second -- weights, AdamW moments and step counts bit-identical (per-epoch seeding, deterministic kernels)."""
    from segland_amd import train_base
    common = ['--model', 'pspnet_pop', '--backbone', 'resnet50', '--dataset', 'synthetic', '--batch-size', '4', '--input-size', '128,128', '--base-size', '128,128',
              '--learning-rate', '1e-4', '--print-frequency', '100', '--num-workers', '0', '--restore-from', '/nonexistent', '--allow-random-init', '--random-seed', '11']
    a, b = str(tmp_path / 'full'), str(tmp_path / 'split')
    train_base.main(common + ['--num-epoch', '2', '--snapshot-dir', a])
    torch.manual_seed(11)                       # the model is built before set_seed(random_seed + epoch): same initial weights for the split run
    train_base.main(common + ['--num-epoch', '1', '--snapshot-dir', b])
    # num-epoch enters the poly LR schedule: continue with the 2-epoch schedule from the state of epoch 1
    st = torch.load(os.path.join(b, 'state_1.pth'), map_location='cpu')
    assert st['epoch'] == 1 and 'optimizer' in st and all(k.startswith('module.') for k in st['state_dict'])
    train_base.main(common + ['--num-epoch', '2', '--snapshot-dir', b, '-c', os.path.join(b, 'state_1.pth')])
    fa, fb = torch.load(os.path.join(a, 'state_2.pth'), map_location='cpu'), torch.load(os.path.join(b, 'state_2.pth'), map_location='cpu')
    # epoch 1 of the split run used lr_poly(epoch 0, max 1) == lr_poly(epoch 0, max 2) == base lr, so both runs saw the same schedule
    for k in fa['state_dict']:
        assert torch.equal(fa['state_dict'][k], fb['state_dict'][k]), k
    sa, sb = fa['optimizer']['state'], fb['optimizer']['state']
    assert sa.keys() == sb.keys() and all(float(sa[i]['step']) == float(sb[i]['step']) and torch.equal(sa[i]['exp_avg'], sb[i]['exp_avg']) for i in sa)
