"""Round 5: the nine-tap weight-gradient kernel (conv_wgrad3.hip) and what else the round added, through the C ABI.
Tolerances as in test_kernels_gpu.py: bf16 kernels 2.5e-2 of the tensor scale (inputs rounded to bf16, fp32 accumulation); run-to-run results bit-identical."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import formula as fm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def rnd(x, dtype):
    return x.to(dtype).float()


def close(got, ref, what, tol=2.5e-2):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    s = float(ref.abs().max())
    err = float((got - ref).abs().max())
    print('%s: max abs err %.3e of scale %.3e' % (what, err, s))
    assert err <= tol * max(s, 1e-6), '%s: max abs err %g vs scale %g' % (what, err, s)


W3_CASES = [
    # B, H, W, Cin, Cout, dil          what the plan makes of it
    (2, 64, 64, 128, 128, 1),        # 8 strips of 64 rows, 2 tiles: strips cut into segments (L + 2 steps per piece)
    (6, 64, 64, 128, 128, 1),        # whole strips
    (2, 64, 64, 256, 256, 2),        # polyphase d = 2: sub-images 32 x 32
    (2, 64, 64, 128, 256, 4),        # d = 4: sub-images 16 x 16, one strip each
    (3, 48, 80, 64, 128, 1),         # five strips per row, Cin = 64 (one c tile)
    (7, 60, 64, 64, 512, 2),         # Hs = 30: a ragged last segment
    (1, 128, 128, 192, 128, 1),      # Cin = 3 tiles of 64
    (2, 50, 96, 128, 128, 2),        # Hs = 25 cut into segments of 13 + 12 rows
]


@pytest.mark.parametrize('case', W3_CASES)
def test_wgrad_nine_tap_kernel(hip, case):
    """3x3 stride-1 layers (pad == dilation) with Cout % 128 == 0 and Cin % 64 == 0 take conv_wgrad3_kernel (resnet.py:46,96-97 backward): against torch's fp32
    gradient, against the per-tap kernels it replaces, bit-stable run to run, and into a wider OIHW tensor at a channel offset."""
    from segland_amd import ops
    B, H, W, Cin, Cout, dil = case
    dtype = torch.bfloat16
    tag = 'w3%s' % (case,)
    x = rnd(fm.sym(tag + 'x', (B, Cin, H, W), 1.0), dtype)
    w = rnd(fm.sym(tag + 'w', (Cout, Cin, 3, 3), (3.0 / (Cin * 9)) ** 0.5), dtype).requires_grad_(True)
    gy = rnd(fm.sym(tag + 'gy', (B, Cout, H, W), 1.0), dtype)
    F.conv2d(x, w, None, 1, dil, dil).backward(gy)
    spec = ops.ConvSpec(Cin, Cout, 3, 1, dil, dil)
    xg, gyg = nhwc(x, dtype), nhwc(gy, dtype)
    d = ops.conv_desc(dtype, B, H, W, spec, None)
    hip.sl_debug_wgrad3(1)
    assert hip.sl_conv2d_wgrad_config(C.byref(d)) == 3, 'the shape must be served by the nine-tap kernel'
    dw = ops.conv2d_bwd_weight(xg, gyg, spec)
    close(dw, w.grad, 'nine-tap wgrad %s vs torch' % (case,))
    assert torch.equal(dw, ops.conv2d_bwd_weight(xg, gyg, spec)), 'run-to-run bit stability'
    hip.sl_debug_wgrad3(0)
    try:
        assert hip.sl_conv2d_wgrad_config(C.byref(d)) != 3
        old = ops.conv2d_bwd_weight(xg, gyg, spec)
    finally:
        hip.sl_debug_wgrad3(1)
    # both sum the same bf16 products in fp32, in different orders
    close(dw, old, 'nine-tap vs per-tap kernel', tol=2e-4)
    wide = torch.full((Cout, Cin + 128, 3, 3), 7.0, device=DEV)
    ops.conv2d_bwd_weight(xg, gyg, spec, out=wide, out_ci_off=64)
    assert torch.equal(wide[:, 64:64 + Cin], dw) and bool((wide[:, :64] == 7).all()) and bool((wide[:, 64 + Cin:] == 7).all())


def test_wgrad_nine_tap_kernel_virtual_concat(hip):
    """x = [x1 | x2] read from two tensors (the pyramid conv's operand form, pspnet_pop.py:33-34): every 64-channel tile lies in one of them."""
    from segland_amd import ops
    dtype = torch.bfloat16
    B, H, W, C1, C2, Cout = 2, 64, 64, 128, 64, 128
    x = rnd(fm.sym('w3cat/x', (B, C1 + C2, H, W), 1.0), dtype)
    w = rnd(fm.sym('w3cat/w', (Cout, C1 + C2, 3, 3), 0.05), dtype).requires_grad_(True)
    gy = rnd(fm.sym('w3cat/gy', (B, Cout, H, W), 1.0), dtype)
    F.conv2d(x, w, None, 1, 1, 1).backward(gy)
    spec = ops.ConvSpec(C1 + C2, Cout, 3, 1, 1, 1)
    dw = ops.conv2d_bwd_weight(nhwc(x[:, :C1], dtype), nhwc(gy, dtype), spec, x2=nhwc(x[:, C1:], dtype))
    close(dw, w.grad, 'nine-tap wgrad, virtual concat')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_stem_pool_backward_emits_bn_backward_statistics(hip, dtype):
    """sl_stem_pool_relu_bwd_bnstat (maxpool + ReLU backward of resnet.py:124-125 with bn1's reduce pass in the same sweep): the gradient is bit-identical to the plain
    kernel's, the column sums equal the sums over that stored gradient (fp64 reference) to 1e-5, and bn_bwd on them equals bn_bwd with its own reduce pass."""
    from segland_amd import ops
    B, Hc, Wc = 3, 64, 96
    c0 = (torch.randn(B, Hc, Wc, 64, device=DEV) * 1.5).to(dtype)
    mean = torch.randn(64, device=DEV) * 0.2
    invstd = torch.rand(64, device=DEV) + 0.5
    gamma = torch.rand(64, device=DEV) + 0.5
    beta = torch.randn(64, device=DEV) * 0.3
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pooled, idx = ops.stem_bn_relu_pool(c0, scale, shift, want_idx=True)
    dp = torch.randn_like(pooled.float()).to(dtype)
    g_plain = ops.stem_pool_relu_bwd(dp, idx, c0, scale, shift)
    g, part = ops.stem_pool_relu_bwd_bnstat(dp, idx, c0, scale, shift, mean, invstd)
    assert torch.equal(g, g_plain)
    g64, x64 = g.double().reshape(-1, 64), c0.double().reshape(-1, 64)
    ref = torch.stack([g64.sum(0), (g64 * ((x64 - mean.double()) * invstd.double())).sum(0)])
    got = part.double().sum(0)
    err = float(((got - ref).abs() / (g64.abs().sum(0) + 1e-9)).max())
    print('stem bnstat column sums: max err relative to sum |g| %.2e' % err)
    assert err < 1e-5
    a = ops.bn_bwd(g, None, c0, mean, invstd, gamma, train=True)
    b = ops.bn_bwd(g, None, c0, mean, invstd, gamma, train=True, pre_partial=part)
    for u, v, what in ((a[0], b[0], 'dx'), (a[2], b[2], 'dgamma'), (a[3], b[3], 'dbeta')):
        close(v, u, 'stem bn1 backward from fused partials: %s' % what, tol=2e-5 if dtype == torch.float32 else 1e-2)


def test_pyramid_stage_batchnorm_backward_in_one_launch(hip):
    """sl_ppm_stage_bn_bwd (BatchNorm + ReLU backward of the four pyramid stages, pspnet_pop.py:12-16, one launch) against ops.bn_bwd level by level (reduce / finalize /
    apply per level): dx, dgamma, dbeta to 1e-5; one level in eval mode (running statistics)."""
    from segland_amd import ops
    B, sizes, Cs = 3, (1, 2, 3, 6), 128
    rows = B * sum(s * s for s in sizes)
    x = torch.randn(rows, Cs, device=DEV)
    dy = torch.randn(rows, Cs, device=DEV)
    means, invs, gammas, ys, off = [], [], [], torch.empty_like(x), 0
    for s in sizes:
        n = B * s * s
        m, v = x[off:off + n].mean(0), x[off:off + n].var(0, unbiased=False)
        i = (v + 1e-5).rsqrt()
        g = torch.rand(Cs, device=DEV) + 0.5
        ys[off:off + n] = torch.relu((x[off:off + n] - m) * i * g + 0.1)
        means.append(m.contiguous()); invs.append(i.contiguous()); gammas.append(g); off += n
    trains = [True, True, False, True]
    dg = [torch.empty(Cs, device=DEV) for _ in sizes]
    db = [torch.empty(Cs, device=DEV) for _ in sizes]
    dx = ops.ppm_stage_bn_bwd(dy, ys, x, B, sizes, means, invs, gammas, trains, dg, db)
    off = 0
    for k, s in enumerate(sizes):
        n = B * s * s
        rdx, _, rdg, rdb = ops.bn_bwd(dy[off:off + n], ys[off:off + n], x[off:off + n], means[k], invs[k], gammas[k], train=trains[k])
        close(dx[off:off + n], rdx, 'level %d dx' % k, tol=1e-5)
        close(dg[k], rdg, 'level %d dgamma' % k, tol=1e-5)
        close(db[k], rdb, 'level %d dbeta' % k, tol=1e-5)
        off += n


@pytest.mark.parametrize('N', [256, 1024, 2048])
def test_k512_layers_on_the_half_tile_kernel(hip, N):
    """1x1 convs with 512 input channels at >= 65 536 pixels (layer4 conv3 forward 512 -> 2048 and the data gradient of layer4 conv1, resnet.py:44,49) run on
    conv_gemm_p8_kernel (round 5's pixel-stationary K = 512 variant measured equal inside the step and was deleted in round 6): forward + BN statistic partials,
    plain data gradient, data gradient + addend, + addend gated by ReLU bits -- against fp32 torch on the same bf16 operands."""
    from segland_amd import ops
    dt_ = torch.bfloat16
    B, H, W, K = 4, 128, 128, 512
    M = B * H * W
    x = torch.randn(B, H, W, K, device=DEV).to(dt_)
    w = (torch.randn(N, K, 1, 1, device=DEV) * (1.0 / K) ** 0.5)
    spec = ops.ConvSpec(K, N, 1, 1, 0, 1)
    wf, _ = ops.weight_prep(w, dt_)
    d = ops.conv_desc(dt_, B, H, W, spec, None)
    assert hip.sl_conv2d_tile_config_ex(C.byref(d), 0, 1) // 1000000 == 5
    y, part = ops.conv2d_fwd(x, wf, spec, want_stats=True)
    ref = x.float().reshape(M, K) @ wf.float().reshape(N, K).t()
    close(y.float().reshape(M, N), ref, 'K = 512 forward N=%d' % N)
    yr = y.float().reshape(M, N)
    s = part.sum(0)
    close(s[0], yr.sum(0), 'K = 512 statistics: sum', tol=1e-4)
    close(s[1], (yr * yr).sum(0), 'K = 512 statistics: sum of squares', tol=1e-4)
    # data gradient of a conv N -> 512 (dy has 512 channels, dx has N)
    spec_b = ops.ConvSpec(N, K, 1, 1, 0, 1)
    wb_src = (torch.randn(K, N, 1, 1, device=DEV) * (1.0 / K) ** 0.5)
    _, wb = ops.weight_prep(wb_src, dt_)
    add = torch.randn(B, H, W, N, device=DEV).to(dt_)
    bits = torch.randint(0, 256, (M * N // 8,), dtype=torch.uint8, device=DEV)
    refd = x.float().reshape(M, K) @ wb.float().reshape(N, K).t()
    bitmask = ((bits.view(-1, 1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(M, N).bool()
    g1 = ops.conv2d_bwd_data(x, wb, spec_b, (H, W))
    close(g1.float().reshape(M, N), refd, 'K = 512 data gradient N=%d' % N)
    g1 = ops.conv2d_bwd_data(x, wb, spec_b, (H, W), addend=add)
    close(g1.float().reshape(M, N), refd.to(dt_).float() + add.float().reshape(M, N), 'K = 512 data gradient + addend N=%d' % N)
    g1 = ops.conv2d_bwd_data(x, wb, spec_b, (H, W), addend=add, addend_mask=bits)
    close(g1.float().reshape(M, N), refd.to(dt_).float() + torch.where(bitmask, add.float().reshape(M, N), torch.zeros((), device=DEV)), 'K = 512 data gradient + gated addend N=%d' % N)


def test_orth_loss_with_aux_preds_golden_g3b(hip):
    """segland_amd.loss.criterion.OrthLoss.forward(aux_preds=...) (criterion.py:56-60: total = seg + 10 orth + 0.4 aux) against golden G3b, which the reference's own
    OrthLoss produced: the four loss values to 2e-5 and the three gradients."""
    import numpy as np
    from conftest import golden
    from segland_amd.loss.criterion import OrthLoss
    g = golden('g3b_loss_aux')
    preds = fm.sym('g3b/preds', (2, 8, 8, 8), 2.0).to(DEV).requires_grad_(True)
    aux = fm.sym('g3b/aux', (2, 8, 16, 16), 1.5).to(DEV).requires_grad_(True)
    target = fm.formula_mask(2, 64, 64, 8, tag='g3b/mask', block=8, ignore_rows=7).to(DEV)
    e = F.normalize(fm.sym('g3b/emb', (7, 512), 1.0), dim=-1)
    sim = (e @ e.t()).to(DEV).requires_grad_(True)
    d = OrthLoss(255)(preds, target, proto_sim=sim, aux_preds=aux)
    assert sorted(d) == ['aux_loss', 'orth_loss', 'seg_loss', 'total_loss']
    d['total_loss'].backward()
    for k, gk in (('total_loss', 'total'), ('seg_loss', 'seg'), ('aux_loss', 'aux'), ('orth_loss', 'orth')):
        np.testing.assert_allclose(float(d[k]), float(g[gk]), rtol=2e-5, err_msg=k)
    for t, gk in ((preds, 'dpreds'), (aux, 'daux'), (sim, 'dsim')):
        np.testing.assert_allclose(t.grad.cpu().numpy(), g[gk], rtol=1e-4, atol=1e-7, err_msg=gk)


@pytest.mark.parametrize('cin,cout', [(1024, 256), (512, 128), (256, 64)])
def test_conv1_dgrad_reduces_for_two_batchnorms_behind_one_relu(hip, cin, cout):
    """resnet.py:71-78 backward into a stage's FIRST bottleneck: its output ReLU sits behind bn3 and the downsample BatchNorm, and the conv1 data gradient of the block
    behind it (pixel-stationary kernel, MODE 5) gates its result with that ReLU's bits and reduces it against BOTH BatchNorm inputs (sl_conv2d_bwd_data_addend_bnstat2):
    gated gradient bit-identical to the single form, first partials identical to it, second partials equal to a reduce pass of their own (1e-5), and
    ops.bn_bwd2 on the pair equals ops.bn_bwd2 with its own dual reduce pass."""
    from segland_amd import _lib, ops
    B, H, W = 16, 64, 64
    if cin == 256:
        H = W = 128
    g = torch.Generator(device='cpu').manual_seed(5)
    spec = ops.ConvSpec(cin, cout, 1, 1, 0, 1)
    _, wb = ops.weight_prep((torch.randn(cout, cin, 1, 1, generator=g) * (3.0 / cin) ** 0.5).to(DEV), torch.bfloat16)
    dy = torch.randn(B, H, W, cout, generator=g).to(torch.bfloat16).to(DEV)
    add = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
    c3 = (torch.randn(B, H, W, cin, generator=g) * 2 + 0.5).to(torch.bfloat16).to(DEV)
    cd = (torch.randn(B, H, W, cin, generator=g) * 1.5 - 0.25).to(torch.bfloat16).to(DEV)
    bits = torch.randint(0, 256, (c3.numel() // 8,), dtype=torch.uint8, generator=g).to(DEV)
    m3, md = torch.randn(cin, generator=g).to(DEV) * 0.3, torch.randn(cin, generator=g).to(DEV) * 0.2
    i3, idd = (torch.rand(cin, generator=g) + 0.5).to(DEV), (torch.rand(cin, generator=g) + 0.7).to(DEV)
    r1 = ops.conv2d_bwd_data_addend_bnstat(dy, wb, spec, (H, W), add, bits, c3, m3, i3)
    r2 = ops.conv2d_bwd_data_addend_bnstat2(dy, wb, spec, (H, W), add, bits, c3, m3, i3, cd, md, idd)
    assert r1 is not None and r2 is not None, 'shape not served'
    assert torch.equal(r2[0], r1[0]) and torch.equal(r2[1], r1[1])
    L = _lib.lib()
    M = B * H * W
    nblk = L.sl_bn_bwd_reduce_rows(M, cin)
    rp = torch.empty((nblk, 2, cin), dtype=torch.float32, device=DEV)
    _lib.check(L.sl_bn_bwd_reduce(ops.dt(cd), ops._p(r1[0]), None, None, ops._p(cd), ops._p(md), ops._p(idd), ops._p(rp), M, cin, ops._s()), 'reduce')
    s_f, s_r = r2[2].double().sum(0), rp.double().sum(0)
    gd = r1[0].double()
    sc1 = float(gd.abs().sum((0, 1, 2)).max())
    sc2 = float((gd * ((cd.double() - md.double()) * idd.double())).abs().sum((0, 1, 2)).max())
    e1, e2 = float((s_f[0] - s_r[0]).abs().max()) / sc1, float((s_f[1] - s_r[1]).abs().max()) / sc2
    print('dual MODE 5 %d -> %d: second BatchNorm column sums rel err %.2e / %.2e' % (cin, cout, e1, e2))
    assert e1 <= 1e-5 and e2 <= 1e-5
    g3, gdd = torch.rand(cin, device=DEV) + 0.5, torch.rand(cin, device=DEV) + 0.5
    ones = torch.full((M * cin // 8,), 255, dtype=torch.uint8, device=DEV)
    a = ops.bn_bwd2(r1[0], ones, c3, m3, i3, g3, cd, md, idd, gdd)
    b = ops.bn_bwd2(r2[0], None, c3, m3, i3, g3, cd, md, idd, gdd, pre_partials=(r2[1], r2[2]))
    for u, v, what in zip(a, b, ('dx3', 'dgamma3', 'dbeta3', 'dxd', 'dgammad', 'dbetad')):
        close(v, u, 'bn_bwd2 from the fused partials: ' + what, tol=1e-2 if what.startswith('dx') else 2e-5)


def test_stage_first_blocks_take_the_dual_route(hip):
    """Model level (R50 bf16, 4 tiles of 256 x 256... the bench shape family): the blocks behind layer1 / layer2 / layer3's first bottleneck hand it both BatchNorms' column
    sums (three sl_conv2d_bwd_data_addend_bnstat2 launches per backward) and every parameter gradient agrees with the route through the dual reduce pass."""
    from segland_amd import functional as sf, ops
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    img = fm.formula_image(16, 512, 512, 'dual/img').to(DEV)
    mask = fm.formula_mask(16, 512, 512, 8, 'dual/mask', block=32, ignore_rows=40).to(DEV)
    torch.manual_seed(3)
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.bfloat16).to(DEV).train()
    calls = [0]
    real = ops.conv2d_bwd_data_addend_bnstat2

    def counted(*a, **k):
        out = real(*a, **k)
        calls[0] += out is not None
        return out
    grads = {}
    try:
        ops.conv2d_bwd_data_addend_bnstat2 = counted
        for flag in (False, True):
            calls[0] = 0
            real_dual = sf._BN_DUAL
            sf._BN_DUAL = real_dual and flag            # False: no link.bnd -> the first blocks run their own dual... (single) reduce passes
            try:
                m.zero_grad(set_to_none=True)
                d = m(img, mask)
                d['total_loss'].backward()
            finally:
                sf._BN_DUAL = real_dual
            grads[flag] = ({k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}, calls[0], float(d['total_loss'].detach()))
    finally:
        ops.conv2d_bwd_data_addend_bnstat2 = real
    (g0, n0, l0), (g1, n1, l1) = grads[False], grads[True]
    num = sum(float(((g1[k] - v) ** 2).sum()) for k, v in g0.items())
    den = sum(float((v ** 2).sum()) for v in g0.values())
    print('dual route: %d launches (off: %d); loss %.6f / %.6f; gradients global relative L2 %.3e' % (n1, n0, l1, l0, (num / den) ** 0.5))
    assert n0 == 0 and n1 == 3 and l0 == l1
    assert (num / den) ** 0.5 <= 2e-2


def test_half_resolution_addend_of_a_stride2_downsample(hip):
    """resnet.py:109-110 / 71-76 backward at a stride-2 stage entry: the downsample branch's data gradient (1x1 stride 2) is non-zero at the even positions only.  It stays
    on its own 64 x 64 grid and enters conv1's data gradient there (sl_conv2d_bwd_data_addend_half): bit-identical to the route through the zero-filled tensor, with and
    without the cross-block statistics; model level: all gradients agree with the scattered route."""
    from segland_amd import functional as sf, ops
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    B, H, W, cin, cout = 16, 128, 128, 256, 128
    g = torch.Generator(device='cpu').manual_seed(9)
    spec = ops.ConvSpec(cin, cout, 1, 1, 0, 1)
    _, wb = ops.weight_prep((torch.randn(cout, cin, 1, 1, generator=g) * (3.0 / cin) ** 0.5).to(DEV), torch.bfloat16)
    dy = torch.randn(B, H, W, cout, generator=g).to(torch.bfloat16).to(DEV)
    half = torch.randn(B, H // 2, W // 2, cin, generator=g).to(torch.bfloat16).to(DEV)
    full = torch.zeros(B, H, W, cin, dtype=torch.bfloat16, device=DEV)
    full[:, ::2, ::2] = half
    x_like = torch.empty(B, H, W, cin, dtype=torch.bfloat16, device=DEV)
    assert ops.conv2d_bwd_data_addend_half_ok(x_like, spec)
    dx, part = ops.conv2d_bwd_data_addend_half(dy, wb, spec, (H, W), half)
    assert part is None and torch.equal(dx, ops.conv2d_bwd_data(dy, wb, spec, (H, W), addend=full))
    c3 = (torch.randn(B, H, W, cin, generator=g) * 2 + 0.5).to(torch.bfloat16).to(DEV)
    bits = torch.randint(0, 256, (c3.numel() // 8,), dtype=torch.uint8, generator=g).to(DEV)
    m3, i3 = torch.randn(cin, generator=g).to(DEV) * 0.3, (torch.rand(cin, generator=g) + 0.5).to(DEV)
    dx2, part2 = ops.conv2d_bwd_data_addend_half(dy, wb, spec, (H, W), half, (bits, c3, m3, i3))
    ref = ops.conv2d_bwd_data_addend_bnstat(dy, wb, spec, (H, W), full, bits, c3, m3, i3)
    assert part2 is not None and torch.equal(dx2, ref[0]) and torch.equal(part2, ref[1])
    # model level
    img = fm.formula_image(16, 512, 512, 'half/img').to(DEV)                 # the pixel-stationary kernel takes layers of >= 65 536 pixels
    mask = fm.formula_mask(16, 512, 512, 8, 'half/mask', block=32, ignore_rows=20).to(DEV)
    torch.manual_seed(3)
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.bfloat16).to(DEV).train()
    calls, real, grads = [0], ops.conv2d_bwd_data_addend_half, {}

    def counted(*a, **k):
        calls[0] += 1
        return real(*a, **k)
    try:
        ops.conv2d_bwd_data_addend_half = counted
        for flag in (False, True):
            sf._DS_HALF, calls[0] = flag, 0
            m.zero_grad(set_to_none=True)
            d = m(img, mask)
            d['total_loss'].backward()
            grads[flag] = ({k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}, calls[0], float(d['total_loss'].detach()))
    finally:
        ops.conv2d_bwd_data_addend_half, sf._DS_HALF = real, True
    (g0, n0, l0), (g1, n1, l1) = grads[False], grads[True]
    num = sum(float(((g1[k] - v) ** 2).sum()) for k, v in g0.items())
    den = sum(float((v ** 2).sum()) for v in g0.values())
    print('half-resolution addend: %d launch(es) (off: %d); loss %.6f / %.6f; gradients global relative L2 %.3e' % (n1, n0, l1, l0, (num / den) ** 0.5))
    assert n0 == 0 and n1 == 1 and l0 == l1 and (num / den) ** 0.5 <= 1e-2


R192_CASES = [
    # B, H, W, K, N, k        Swin-T stage 2 on a 512 x 512 tile: 64 x 64 tokens, 192 channels (swintransformer.py:127-140, 249-252)
    (8, 64, 64, 768, 192, 1),      # Mlp fc2
    (2, 64, 64, 192, 576, 1),      # qkv: three 192-column tiles
    (1, 100, 60, 192, 192, 1),     # proj on a ragged row count (6 000 rows: a partial last row tile)
    (2, 32, 48, 192, 192, 3),      # a 3x3 gather with 192 output channels
]


@pytest.mark.parametrize('case', R192_CASES)
def test_192_column_ring_tile(hip, case):
    """Output widths that are 64- but not 128-multiples (192, 576: Swin-T stage 2) run on 128 x 192 tiles of the 4-stage ring kernel instead of the two-stage 256 x 64
    kernel: forward (+ BN statistic partials), the Linear epilogue (bias, DropPath row scale, residual, GELU side output) and the data gradient (+ addend), against fp32
    torch and, for the 1x1 layers, bit-identical to the kernel they ran on before (same MFMA sequence per output element)."""
    from segland_amd import ops
    dt_ = torch.bfloat16
    B, H, W, K, N, k = case
    M = B * H * W
    torch.manual_seed(5)
    x = torch.randn(B, H, W, K, device=DEV).to(dt_)
    w = (torch.randn(N, K, k, k, device=DEV) * (1.0 / (K * k * k)) ** 0.5)
    spec = ops.ConvSpec(K, N, k, 1, k // 2, 1)
    wf, wb = ops.weight_prep(w, dt_)
    d = ops.conv_desc(dt_, B, H, W, spec, None)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dt_).float(), padding=k // 2).permute(0, 2, 3, 1).reshape(M, N)

    def both(fn):
        hip.sl_debug_conv_ring192(1)
        assert hip.sl_conv2d_tile_config_ex(C.byref(d), 0, 0) == 4128192
        a = fn()
        hip.sl_debug_conv_ring192(0)
        try:
            assert hip.sl_conv2d_tile_config_ex(C.byref(d), 0, 0) != 4128192
            b = fn()
        finally:
            hip.sl_debug_conv_ring192(1)
        return a, b

    (y, part), (y0, part0) = both(lambda: ops.conv2d_fwd(x, wf, spec, want_stats=True))
    close(y.float().reshape(M, N), ref, '192-column tile forward %s' % (case,))
    if k == 1:          # a 3x3 gather walks (channel chunk, tap) in another order on the ring: same sums, other rounding
        assert torch.equal(y, y0), 'forward: 128 x 192 ring tile vs two-stage kernel'
    else:
        close(y, y0, 'forward vs two-stage kernel', tol=1e-2)
    yr = y.float().reshape(M, N)
    close(part.sum(0)[0], yr.sum(0), 'statistics: sum', tol=1e-4)
    close(part.sum(0)[1], (yr * yr).sum(0), 'statistics: sum of squares', tol=1e-4)
    if k == 1:
        bias = torch.randn(N, device=DEV)
        rs = torch.rand(B, device=DEV) + 0.5
        res = torch.randn(B, H, W, N, device=DEV).to(dt_)
        (l1, g1), (l0, g0) = both(lambda: ops.linear_fwd(x, wf, spec, bias=bias, want_gelu=True))
        close(l1.float().reshape(M, N), ref + bias, 'Linear + bias')
        close(g1.float().reshape(M, N), F.gelu(l1.float().reshape(M, N)), 'GELU side output', tol=1e-2)
        assert torch.equal(l1, l0) and torch.equal(g1, g0)
        l1, l0 = both(lambda: ops.linear_fwd(x, wf, spec, bias=bias, row_scale=rs, residual=res))
        want = (ref + bias).reshape(B, H * W, N) * rs.view(B, 1, 1) + res.float().reshape(B, H * W, N)
        close(l1.float().reshape(B, H * W, N), want, 'Linear + DropPath scale + residual')
        assert torch.equal(l1, l0)
    # data gradient of a conv N -> K' with K' = 192-multiple: dy has Kd channels, dx has Nd = 192 / 576
    Kd, Nd = (K, N)
    spec_b = ops.ConvSpec(Nd, Kd, k, 1, k // 2, 1)
    w2 = (torch.randn(Kd, Nd, k, k, device=DEV) * (1.0 / (Kd * k * k)) ** 0.5)
    _, wb2 = ops.weight_prep(w2, dt_)
    refd = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w2.to(dt_).float(), padding=k // 2).permute(0, 2, 3, 1).reshape(M, Nd)
    add = torch.randn(B, H, W, Nd, device=DEV).to(dt_)
    db = ops.conv_desc(dt_, B, H, W, spec_b, None)
    hip.sl_debug_conv_ring192(1)
    assert hip.sl_conv2d_tile_config_ex(C.byref(db), 1, 0) == 4128192
    g1 = ops.conv2d_bwd_data(x, wb2, spec_b, (H, W))
    ga1 = ops.conv2d_bwd_data(x, wb2, spec_b, (H, W), addend=add)
    hip.sl_debug_conv_ring192(0)
    try:
        g0 = ops.conv2d_bwd_data(x, wb2, spec_b, (H, W))
        ga0 = ops.conv2d_bwd_data(x, wb2, spec_b, (H, W), addend=add)
    finally:
        hip.sl_debug_conv_ring192(1)
    close(g1.float().reshape(M, Nd), refd, 'data gradient')
    close(ga1.float().reshape(M, Nd), refd.to(dt_).float() + add.float().reshape(M, Nd), 'data gradient + addend')
    if k == 1:
        assert torch.equal(g1, g0) and torch.equal(ga1, ga0), 'data gradient: 128 x 192 ring tile vs two-stage kernel'
    else:
        close(g1, g0, 'data gradient vs two-stage kernel', tol=1e-2)


@pytest.mark.parametrize('case', [(2, 64, 64, 192, 768), (1, 64, 64, 128, 384), (1, 32, 32, 384, 1536), (1, 16, 16, 768, 3072), (1, 50, 52, 128, 384), (3, 20, 28, 192, 768)])   # the last two: a partial last row tile
def test_data_gradient_behind_a_gelu(hip, case):
    """Mlp backward (swintransformer.py:26-31): fc2's data gradient times GELU'(h) in the GEMM's store phase (sl_conv2d_bwd_data_gelu) equals the data gradient
    followed by sl_gelu_bwd bit for bit (the store phase works on the rounded data gradient), on every kernel the Swin-T stages dispatch to, and is torch's."""
    from segland_amd import ops, ops_swin as osw
    dt_ = torch.bfloat16
    B, H, W, Cn, Hid = case
    M = B * H * W
    torch.manual_seed(7)
    dz = torch.randn(B, H, W, Cn, device=DEV).to(dt_)                     # gradient behind fc2 (Hid -> Cn)
    h = (torch.randn(B, H, W, Hid, device=DEV) * 1.5).to(dt_)              # fc1's stored pre-activation
    w2 = torch.randn(Cn, Hid, 1, 1, device=DEV) * (1.0 / Hid) ** 0.5
    spec = ops.ConvSpec(Hid, Cn, 1, 1, 0, 1)
    _, wb = ops.weight_prep(w2, dt_)
    d = ops.conv_desc(dt_, B, H, W, spec, None)
    fam = hip.sl_conv2d_tile_config_ex(C.byref(d), 1, 64) // 1000000
    assert fam in (4, 2), 'a tile kernel (the persistent kernels have no GELU store phase): %d' % fam
    fused = ops.conv2d_bwd_data_gelu(dz, wb, spec, (H, W), h)
    two = osw.gelu_bwd(h, ops.conv2d_bwd_data(dz, wb, spec, (H, W)))
    assert torch.equal(fused, two), 'fused vs data gradient + gelu_bwd'
    hf = h.float().reshape(M, Hid).requires_grad_(True)
    y = F.gelu(hf) @ w2.to(dt_).float().reshape(Cn, Hid).t()
    y.backward(dz.float().reshape(M, Cn))
    close(fused.float().reshape(M, Hid), hf.grad, 'data gradient behind the GELU %s' % (case,))


@pytest.mark.parametrize('case', [
    # tokens (B, H, W), Cin, Cout, dtype                  tile of the weight-gradient kernel
    ((8, 128, 128), 128, 384, torch.bfloat16),          # 128 x 128 (Swin-T stage 1 qkv / fc1 at pitch 128)
    ((2, 64, 64), 768, 192, torch.bfloat16),            # pixel pairs (192 is not a 128-multiple): two partial rows per split
    ((2, 64, 64), 256, 384, torch.bfloat16),            # 128 x 256
    ((1, 64, 64), 384, 1536, torch.bfloat16),           # 256 x 128
    ((1, 32, 32), 768, 3072, torch.bfloat16),           # 256 x 256: no bias instantiation, the column sums stay in the reduce launch
    ((1, 40, 52), 128, 384, torch.float32),             # fp32, ragged row count
])
def test_bias_gradient_inside_the_weight_gradient_kernel(hip, case):
    """nn.Linear backward (swintransformer.py:26-31, 95-98): db = sum_rows dy from the weight-gradient kernel's own dy fragments (times an all-ones fragment on the
    MFMA) instead of a second pass over dy: equals the column sums, dw is bit-identical to the launch without it (hook sl_debug_wgrad_bias(0))."""
    from segland_amd import ops
    (B, H, W), K, N, dt_ = case
    M = B * H * W
    torch.manual_seed(11)
    x = torch.randn(B, H, W, K, device=DEV).to(dt_)
    dy = (torch.randn(B, H, W, N, device=DEV) + 0.25).to(dt_)
    spec = ops.ConvSpec(K, N, 1, 1, 0, 1)
    d = ops.conv_desc(dt_, B, H, W, spec, None)
    rows1 = hip.sl_conv2d_bwd_weight_bias_rows(C.byref(d), 0, 0)
    dw1, db1 = ops.conv2d_bwd_weight_bias(x, dy, spec)
    hip.sl_debug_wgrad_bias(0)
    try:
        rows0 = hip.sl_conv2d_bwd_weight_bias_rows(C.byref(d), 0, 0)
        dw0, db0 = ops.conv2d_bwd_weight_bias(x, dy, spec)
    finally:
        hip.sl_debug_wgrad_bias(1)
    print('partial rows: %d in the kernel, %d by the column-sum blocks' % (rows1, rows0))
    assert torch.equal(dw1, dw0), 'weight gradient with / without the bias fragment'
    ref = dy.float().reshape(M, N).sum(0)
    close(db1, ref, 'bias gradient %s' % (case,), tol=2e-5 if dt_ == torch.float32 else 1e-5)
    close(db0, ref, 'bias gradient (reduce launch)', tol=2e-5 if dt_ == torch.float32 else 1e-5)
    close(dw1.reshape(N, K), dy.float().reshape(M, N).t() @ x.float().reshape(M, K), 'weight gradient', tol=2.5e-2 if dt_ == torch.bfloat16 else 1e-4)
    # zero-padded channel counts (Swin-T stage 1: C = 96 at pitch 128): the parameter-shaped outputs
    if K == 128 and dt_ == torch.bfloat16:
        dwc, dbc = ops.conv2d_bwd_weight_clip(x, dy, spec, N - 96, 96, want_bias=True)
        assert torch.equal(dwc, dw1[:N - 96, :96]) and torch.equal(dbc, db1)


@pytest.mark.parametrize('case', [(14, 512, 512), (8, 512, 512), (30, 512, 512), (32, 2048, 64), (1, 64, 128)])
def test_prototype_rows_kernel(hip, case):
    """The +-prototype rows of the POP head's classifier MLP (pspnet_pop.py:46-52 on [2K, 512]; functional._row_parts) are launches of <= 32 rows: conv_rows_small_kernel
    (one wave per 32 columns, operands straight from global memory) against the tile kernel they ran on (hook off): bit-identical forward + ReLU and data gradient + ReLU
    mask, and against fp32 torch."""
    from segland_amd import ops
    dt_ = torch.bfloat16
    M, K, N = case
    torch.manual_seed(M + K)
    x = torch.randn(1, 1, M, K, device=DEV).to(dt_)
    w = torch.randn(N, K, 1, 1, device=DEV) * (1.0 / K) ** 0.5
    spec = ops.ConvSpec(K, N, 1, 1, 0, 1)
    wf, wb = ops.weight_prep(w, dt_)
    d = ops.conv_desc(dt_, 1, 1, M, spec, None)
    assert hip.sl_conv2d_tile_config_ex(C.byref(d), 0, 0) == 3032032
    y1 = ops.conv2d_fwd(x, wf, spec, relu=True)[0]
    dy = torch.randn(1, 1, M, N, device=DEV).to(dt_)
    hm = torch.randn(1, 1, M, K, device=DEV).to(dt_)
    g1 = ops.conv2d_bwd_data(dy, wb, spec, (1, M), mask_src=hm)
    p1 = ops.conv2d_bwd_data(dy, wb, spec, (1, M))
    hip.sl_debug_conv_rows_small(0)
    try:
        assert hip.sl_conv2d_tile_config_ex(C.byref(d), 0, 0) != 3032032
        y0 = ops.conv2d_fwd(x, wf, spec, relu=True)[0]
        g0 = ops.conv2d_bwd_data(dy, wb, spec, (1, M), mask_src=hm)
        p0 = ops.conv2d_bwd_data(dy, wb, spec, (1, M))
    finally:
        hip.sl_debug_conv_rows_small(1)
    assert torch.equal(y1, y0) and torch.equal(g1, g0) and torch.equal(p1, p0), 'prototype-rows kernel vs tile kernel'
    ref = torch.relu(x.float().reshape(M, K) @ w.to(dt_).float().reshape(N, K).t())
    close(y1.float().reshape(M, N), ref, 'forward + ReLU %s' % (case,))
    refg = dy.float().reshape(M, N) @ w.to(dt_).float().reshape(N, K)
    close(p1.float().reshape(M, K), refg, 'data gradient')
    close(g1.float().reshape(M, K), torch.where(hm.float().reshape(M, K) > 0, refg.to(dt_).float(), torch.zeros((), device=DEV)), 'data gradient + ReLU mask')
