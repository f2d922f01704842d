"""Round 6: the bf16 DEFAULT dispatch is what the parity gates run, the fine-tune head's frozen base chain really is cached (advisor), and the
kernels the round added.  Tolerances as in test_kernels_gpu.py / test_round5_gpu.py: bf16 kernels 2.5e-2 of the tensor scale, gradients in relative L2."""
import ctypes as C

import pytest
import torch
import torch.nn as nn

from oracle import formula as fm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def nchw(x):
    return x.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.detach().float().cpu().double()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


# The kernels `python bench.py` runs the ResNet-50 step on (profiles/r5_conv_shapes.txt): per bench-sized block, the families the instrumented launches must be
# attributed to.  A dispatch change that silently routes these shapes to a fallback kernel fails here.
BENCH_BLOCKS = {
    # name: (inplanes, planes, dilation, B, H, W, families that MUST appear)
    'layer3': (1024, 256, 2, 16, 64, 64, {'conv_gemm_p8_kernel<bf16, 256, 256>', 'conv_gemm_p9_kernel<bf16, 256, 256>', 'conv_gemm_sk_kernel<bf16, 256, 64>', 'conv_wgrad3_kernel',
                                          'conv_wgrad_glds_kernel<bf16, 128, 256>'}),
    'layer1': (256, 64, 1, 4, 128, 128, {'conv_gemm_sk_kernel<bf16, 256, 64>', 'conv_c64k3_kernel<bf16, 16, 16>', 'conv_wgrad_c64k3_kernel', 'conv_wgrad_c64p_kernel',
                                         'conv_gemm_glds_kernel<bf16, 256, 64>'}),
    'layer4': (2048, 512, 4, 16, 64, 64, {'conv_gemm_p8_kernel<bf16, 256, 256>', 'conv_gemm_p9_kernel<bf16, 256, 256>', 'conv_wgrad3_kernel', 'conv_wgrad_glds_kernel<bf16, 256, 256>'}),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize('name', list(BENCH_BLOCKS))
def test_bench_sized_bottleneck_pair_bf16_default_dispatch_vs_oracle(hip, name):
    """Two consecutive identity bottlenecks (resnet.py:57-78) at the BENCH shapes of layer1 / layer3 / layer4 (the G5 goldens use 16 x 16 maps, which dispatch to the
    small-map kernels), bf16, train-mode BatchNorm, through the default dispatch: (1) the instrumented launches are attributed to the kernels the bench runs --
    half-tile, 3x3 patch, pixel-stationary, nine-tap weight gradient, ...; (2) the cross-block BatchNorm fusions are taken (ONE stand-alone bn_bwd_reduce launch for six
    BatchNorms: the last block's bn3); (3) output, input gradient and every weight gradient agree with the fp32 CPU oracle on this machine at the bf16 gates."""
    from oracle import pop_oracle as po
    from segland_amd import ops
    from segland_amd.functional import flush_num_batches_tracked
    from segland_amd.networks.backbones.resnet import Bottleneck
    inp, pl, dil, B, H, W, want = BENCH_BLOCKS[name]
    torch.manual_seed(11)
    blocks = [Bottleneck(inp, pl, stride=1, dilation=dil) for _ in range(2)]
    oracles = [po.make_bottleneck(inp, pl, 1, dil, False) for _ in range(2)]
    for b, o in zip(blocks, oracles):
        with torch.no_grad():
            for m in b.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.copy_(m.weight.to(torch.bfloat16).float())        # both sides multiply the same bf16 weights
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
        o.load_state_dict(b.state_dict())
        b.to(DEV).train(); o.train()
    blocks[1].__dict__['_sl_prev'] = blocks[0]          # what ResNet.base_forward does (resnet.py of the build): block 1's only input is block 0's output
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, inp, H, W, generator=g).relu_().to(torch.bfloat16).float()
    coef = torch.randn(B, inp, H, W, generator=g)
    xg = nhwc(x, torch.bfloat16).requires_grad_(True)
    ops.PROFILER.start()
    y = blocks[1](blocks[0](xg))
    (y.float() * nhwc(coef, torch.float32)).sum().backward()
    table = ops.PROFILER.stop()
    tb = ops.PROFILER.stop_bytes()
    flush_num_batches_tracked()
    seen = set(table)
    print('%s pair: conv families %s; bn_bwd_reduce launches %d' % (name, sorted(seen), tb.get('bn_bwd_reduce', {}).get('calls', 0)))
    assert want <= seen, 'bench kernels not taken: %s' % sorted(want - seen)
    assert not any(f.startswith('conv_wgrad_kernel') or f.startswith('conv_gemm_kernel') for f in seen), seen       # the register-staged fallbacks
    # layer4's conv1 data gradients (K = 512) run on the half-tile kernel, whose store phase does not carry the cross-block sums: both bn3 passes stand alone
    assert tb.get('bn_bwd_reduce', {}).get('calls', 0) == (2 if name == 'layer4' else 1), tb.get('bn_bwd_reduce')
    xo = x.clone().requires_grad_(True)
    yo = po.bottleneck_forward(oracles[1], po.bottleneck_forward(oracles[0], xo))
    (yo * coef).sum().backward()
    e = rel_l2(nchw(y), yo)
    print('  y rel L2 %.4f' % e)
    assert e <= 2e-2, e
    # gradients: the bf16 gate of test_model_gpu.py (GTOLS: relative L2 0.15 -- every stored activation and gradient of the twelve layers is rounded to 8 mantissa bits and
    # ReLU-mask flips move whole elements) plus a cosine
    def cos(a, b):
        a, b = a.detach().float().cpu().double().reshape(-1), b.detach().float().cpu().double().reshape(-1)
        return float(a @ b / (a.norm() * b.norm()))
    e, c = rel_l2(nchw(xg.grad), xo.grad), cos(nchw(xg.grad), xo.grad)
    print('  dx rel L2 %.4f cosine %.5f' % (e, c))
    assert e <= 0.15 and c >= 0.985, (e, c)
    for bi, (b, o) in enumerate(zip(blocks, oracles)):
        for k, p in b.named_parameters():
            ref = dict(o.named_parameters())[k].grad
            e, c = rel_l2(p.grad, ref), cos(p.grad, ref)
            print('  block %d d_%s rel L2 %.4f cosine %.5f' % (bi, k, e, c))
            assert e <= 0.15 and c >= 0.985, (bi, k, e, c)


def test_dispatch_table_of_the_bench_shapes(hip):
    """Host logic only (sl_conv2d_tile_config_ex / sl_conv2d_wgrad_config: the same predicate chain the launches switch on): the ResNet-50 bench shapes and the kernel
    family each must run on.  Family codes: 8 = 3x3 patch, 6 = pixel-stationary, 5 = half-tile, 7 = 64 -> 64 patch; weight gradients 3 = nine-tap, 1 / 2 = the 64-channel
    kernels, 10256256 / 10128256 = the LDS-DMA tile kernel."""
    from segland_amd import _lib
    from segland_amd.ops import EPI_ADDEND, EPI_GATE, EPI_STATS
    bf = _lib.SL_BF16

    def desc(B, H, W, cin, cout, k, dil=1, stride=1):
        pad = dil if k == 3 else 0
        Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        return _lib.SlConvDesc(bf, B, H, W, cin, cout, k, k, stride, pad, dil, Ho, Wo, cin)
    fam = lambda d, mode, epi: hip.sl_conv2d_tile_config_ex(C.byref(d), mode, epi) // 1000000 % 10      # noqa: E731
    for cin, cout, dil in ((256, 256, 2), (512, 512, 4), (2048, 512, 1)):                                  # layer3 / layer4 conv2, the pyramid conv
        d = desc(16, 64, 64, cin, cout, 3, dil)
        assert fam(d, 0, EPI_STATS) == 8 and fam(d, 1, 0) == 8 and fam(d, 1, EPI_GATE) == 8, (cin, cout)
        assert hip.sl_conv2d_wgrad_config(C.byref(d)) == 3, (cin, cout)
    d = desc(16, 64, 64, 1024, 256, 1)                     # layer3 conv1: forward K = 1024 -> half-tile; data gradient K = 256 -> pixel-stationary with the cross-block store loop
    assert fam(d, 0, EPI_STATS) == 5 and fam(d, 1, EPI_GATE | EPI_ADDEND) == 6 and hip.sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(d)) == 256
    assert hip.sl_conv2d_wgrad_config(C.byref(d)) == 10128256
    d = desc(16, 64, 64, 256, 1024, 1)                     # layer3 conv3: forward K = 256 -> pixel-stationary; data gradient K = 1024 -> half-tile with gated statistics
    assert fam(d, 0, EPI_STATS) == 6 and fam(d, 1, EPI_GATE) == 5 and hip.sl_conv2d_bwd_data_bnstat_rows(C.byref(d)) == 256
    d = desc(16, 64, 64, 512, 2048, 1)                     # layer4 conv3
    assert fam(d, 0, EPI_STATS) == 5 and fam(d, 1, EPI_GATE) == 5 and hip.sl_conv2d_wgrad_config(C.byref(d)) == 10256256
    d = desc(16, 128, 128, 64, 64, 3)                      # layer1 conv2
    assert fam(d, 0, EPI_STATS) == 7 and hip.sl_conv2d_wgrad_config(C.byref(d)) == 1
    d = desc(16, 128, 128, 64, 256, 1)                     # layer1 conv3
    assert fam(d, 0, EPI_STATS) == 6 and hip.sl_conv2d_wgrad_config(C.byref(d)) == 2


def test_ft_base_chain_cache_engages_and_is_part_of_the_graph_key(hip):
    """Round-5 advisor: every output of an autograd.Function requires grad when any input does, so the frozen base prototypes of ft mode looked trainable to the head
    and functional._base_chain never cached.  ProtoFn now marks them non-differentiable: after ONE fine-tune training step the cache entry exists, a second step reuses
    it (same tensors), and what it is computed from is part of GraphedStep's state key (a base-classifier weight change behind a captured step re-captures)."""
    from segland_amd import functional as sf
    from segland_amd import graph_step
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    torch.manual_seed(0)
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), is_ft=True, n_novel=4, backbone='resnet50', pretrained_model=None, compute_dtype=torch.bfloat16, dilated=True, os=8).to(DEV)
    m.init_cls_n()
    m.train_mode()
    m.ft_freeze()
    g = torch.Generator().manual_seed(7)
    img, img_b = torch.randn(1, 3, 128, 128, generator=g).to(DEV), torch.randn(1, 3, 128, 128, generator=g).to(DEV)
    mask = torch.randint(8, 12, (1, 128, 128), generator=g).to(DEV)
    mask_b = torch.randint(0, 8, (1, 128, 128), generator=g).to(DEV)
    assert '_sl_base_chain' not in m.__dict__
    m(img, mask, img_b, mask_b.clone())['total_loss'].backward()
    ent = m.__dict__.get('_sl_base_chain')
    assert ent is not None, 'the frozen base chain was not cached'
    assert m.base_emb.grad is None and all(p.grad is None for p in m.classifier.parameters())
    m(img, mask, img_b, mask_b.clone())['total_loss'].backward()
    assert m.__dict__['_sl_base_chain'][1][0] is ent[1][0], 'second step recomputed the cached rows'
    gs = graph_step.GraphedStep(lambda *a: None, m)
    k0 = gs._state_key((img,))
    with torch.no_grad():
        m.classifier[0].weight.add_(0.0)                  # an in-place write (what load_state_dict does): the version counter moves
    assert gs._state_key((img,)) != k0 and sf.base_chain_key(m) is not None


@pytest.mark.parametrize('cin,cout,B,H', [(128, 128, 16, 128), (256, 256, 16, 128), (128, 256, 24, 64)])
def test_stride2_data_gradient_as_parity_planes(hip, cin, cout, B, H):
    """The data gradient of a stride-2 3x3 conv (resnet.py:46: a stage entry's conv2; the bench has one, 128 -> 128 at 128 x 128) runs as four parity-plane launches --
    each destination parity class only walks the taps that reach it (4 / 2 / 2 / 1 of 9): against torch's fp32 gradient on the same bf16 operands, bit-identical to the
    one-launch route it replaces (hook sl_debug_conv_parity(0): the same products in the same order, the skipped ones were exact zeros), plain and with the gated
    BatchNorm-backward statistics in the store phase (column sums of the two routes to 1e-5); and faster."""
    import torch.nn.functional as F
    from segland_amd import ops
    dt_ = torch.bfloat16
    torch.manual_seed(5)
    spec = ops.ConvSpec(cin, cout, 3, 2, 1, 1)
    Ho = H // 2
    w = torch.randn(cout, cin, 3, 3, device=DEV) * (1.0 / (9 * cin)) ** 0.5
    _, wb = ops.weight_prep(w, dt_)
    dy = torch.randn(B, Ho, Ho, cout, device=DEV).to(dt_)
    x0 = torch.zeros(B, cin, H, H, device=DEV, requires_grad=True)
    (F.conv2d(x0, w.to(dt_).float(), stride=2, padding=1) * dy.float().permute(0, 3, 1, 2)).sum().backward()
    ref = x0.grad.permute(0, 2, 3, 1).contiguous()

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        return out, e0.elapsed_time(e1) / 10
    g1, t1 = timed(lambda: ops.conv2d_bwd_data(dy, wb, spec, (H, H)))
    hip.sl_debug_conv_parity(0)
    try:
        g0, t0 = timed(lambda: ops.conv2d_bwd_data(dy, wb, spec, (H, H)))
    finally:
        hip.sl_debug_conv_parity(1)
    s = float(ref.abs().max())
    err = float((g1.float() - ref).abs().max())
    print('stride-2 3x3 data gradient %d -> %d, B %d %dx%d: parity planes %.1f us, one launch %.1f us; max err %.2e of scale %.2e' % (cin, cout, B, H, H, 1e3 * t1, 1e3 * t0, err, s))
    assert err <= 2.5e-2 * s
    assert torch.equal(g1, g0), 'parity planes vs one launch: not bit-identical'
    # gated + BatchNorm-backward column sums of the layer below (MODE 3 of the store phase)
    bn_x = torch.randn(B, H, H, cin, device=DEV).to(dt_)
    gate = torch.randint(0, 256, (B * H * H * cin // 8,), dtype=torch.uint8, device=DEV)
    mean, invstd = torch.randn(cin, device=DEV) * 0.1, torch.rand(cin, device=DEV) + 0.5
    r1 = ops.conv2d_bwd_data_bnstat(dy, wb, spec, (H, H), gate, bn_x, mean, invstd)
    assert r1 is not None
    hip.sl_debug_conv_parity(0)
    try:
        r0 = ops.conv2d_bwd_data_bnstat(dy, wb, spec, (H, H), gate, bn_x, mean, invstd)
    finally:
        hip.sl_debug_conv_parity(1)
    assert torch.equal(r1[0], r0[0]), 'gated gradient: parity planes vs one launch'
    bits = ((gate.view(-1, 1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(B, H, H, cin).bool()
    assert torch.equal(r1[0], torch.where(bits, g1, torch.zeros((), dtype=dt_, device=DEV)))
    s1, s0 = r1[1].sum(0), r0[1].sum(0)
    rel = float((s1 - s0).abs().max() / s0.abs().max())
    print('  gated statistics: column sums of the two routes differ by %.1e of scale' % rel)
    assert rel <= 1e-5
    assert t1 < t0, (t1, t0)


def test_ppm_rows_weight_gradients_grouped(hip):
    """The weight gradients of the pyramid's row GEMMs (four stage convs, pspnet_pop.py:12-16, and the four per-level GEMMs of the factorised prior path) as ONE launch each
    (sl_ppm_rows_wgrad): against torch's fp32 einsum per level (1e-5 of scale: exact-fp32 MFMA products, ascending-row sums), into caller-provided destinations, and
    at the model level -- every PSPModule parameter gradient of the grouped route against the per-level route it replaces."""
    from segland_amd import functional as sf
    from segland_amd import ops
    from segland_amd.networks.pspnet_pop import PSPModule
    torch.manual_seed(2)
    B, sizes = 3, (1, 2, 3, 6)
    rows = ops.ppm_rows(B, sizes)
    for N, K in ((128, 64), (512, 2048), (576, 128)):
        a, x = torch.randn(rows, N, device=DEV), torch.randn(rows, K, device=DEV)
        dst = torch.zeros(N * K, device=DEV)
        dws = ops.ppm_rows_wgrad(a, x, B, sizes, outs=[None, dst, None, None])
        assert dws[1] is dst
        off = 0
        for l, s in enumerate(sizes):
            n = B * s * s
            ref = a[off:off + n].double().t() @ x[off:off + n].double()
            err = float((dws[l].view(N, K).double() - ref).abs().max() / ref.abs().max())
            assert err <= 1e-5, (N, K, l, err)
            off += n
    outs = {}
    for grouped in (True, False):
        old = sf._PPM_WGRAD_GROUPED
        sf._PPM_WGRAD_GROUPED = grouped
        try:
            torch.manual_seed(4)
            dec = PSPModule(256, out_features=128).to(DEV).train()
            xg = torch.randn(2, 24, 24, 256, device=DEV).to(torch.bfloat16).requires_grad_(True)
            y = dec(xg)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum().backward()
            outs[grouped] = {k: p.grad.clone() for k, p in dec.named_parameters()} | {'x': xg.grad.float().clone()}
        finally:
            sf._PPM_WGRAD_GROUPED = old
    for k in outs[True]:
        e = rel_l2(outs[True][k], outs[False][k])
        assert e <= 1e-4, (k, e)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('B,H,W,N', [(16, 64, 64, 512), (3, 24, 40, 128), (2, 7, 9, 64)])
def test_ppm_factorised_scatter_gather_sliding_window(hip, B, H, W, N, dtype):
    """The factorised prior path's scatter (the backward of pspnet_pop.py:19 applied to the bilinearly upsampled stage maps, ppm.hip) as sliding-window kernels: every input
    element read once.  Bit-identical to the general two-stage kernels they replace (hook sl_debug_ppm_fact_walk(0): the same products summed in the same ascending order),
    and the scatter is the exact transpose of the gather (<gather(q), d> == <q, scatter(d)> to fp32 accuracy).  (The gather keeps its general kernels: the hook changes
    nothing there, the comparison stays as a guard.)"""
    from segland_amd import ops
    torch.manual_seed(9)
    sizes = (1, 2, 3, 6)
    rows = ops.ppm_rows(B, sizes)
    q = torch.randn(rows, 9 * N, device=DEV)
    dcb = torch.randn(B, H, W, N, device=DEV).to(dtype)
    shape = (B, H, W, 2048)

    def both(fn):
        r1 = fn()
        hip.sl_debug_ppm_fact_walk(0)
        try:
            r0 = fn()
        finally:
            hip.sl_debug_ppm_fact_walk(1)
        return r1, r0
    s1, s0 = both(lambda: ops.ppm_fact_scatter(dcb, shape, sizes))
    assert torch.equal(s1, s0), 'scatter: sliding window vs general kernels: max diff %g' % float((s1 - s0).abs().max())
    g1, g0 = both(lambda: ops.ppm_fact_gather(q, shape, sizes, N, dtype))
    assert torch.equal(g1, g0), 'gather: sliding window vs general kernels: max diff %g' % float((g1.float() - g0.float()).abs().max())
    if dtype == torch.float32:
        lhs = float((g1.double() * dcb.double()).sum())
        rhs = float((q.double() * s1.double()).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


@pytest.mark.parametrize('dim,heads,H,W', [(96, 3, 14, 21), (192, 6, 14, 14)])
def test_swin_stage_tail_fusions_are_bit_identical(hip, dim, heads, H, W):
    """Round-6 launch fusions of the Swin block backward: (a) DropPath's per-sample factor on a branch's incoming gradient written by the LayerNorm backward that produced the
    gradient (sl_layernorm_bwd_scaled; across blocks through functional_swin.SwinLink), (b) the four nn.Linear slab reduces + the bias column sums that ride in them in
    one launch (sl_conv2d_bwd_weight_defer / sl_wgrad_reduce_multi), (c) the pad tokens' share of d qkv.bias added inside the relative-position table launch
    (sl_relpos_table_grad_bias).  All promise the bits of the launches they replace: two DropPath blocks, every gradient torch.equal."""
    from segland_amd import functional_swin as fs
    from segland_amd.networks.backbones.swintransformer import BasicLayer
    from segland_amd.ops_swin import pad_to
    torch.manual_seed(11)
    st = BasicLayer(dim, 2, heads, [0.1, 0.1], 0, False).to(DEV)
    P = pad_to(dim)
    B = 3
    x0 = torch.zeros(B, H, W, P, device=DEV)
    x0[..., :dim] = torch.randn(B, H, W, dim, device=DEV)
    x0 = x0.to(torch.bfloat16)
    sc = [torch.tensor(v, device=DEV) for v in ([0.0, 1 / 0.9, 1 / 0.9], [1 / 0.9, 0.0, 1 / 0.9], [1 / 0.9, 1 / 0.9, 0.0], [1 / 0.9, 1 / 0.9, 1 / 0.9])]
    wgt = torch.linspace(-1, 1, B * H * W * dim, device=DEV).view(B, H, W, dim)

    def run(ln_scale, wbatch, tail=False):
        old = fs._LN_SCALE, fs._WGRAD_BATCH, fs._BIAS_TAIL
        fs._LN_SCALE, fs._WGRAD_BATCH, fs._BIAS_TAIL = ln_scale, wbatch, tail
        try:
            st.zero_grad(set_to_none=True)
            xg = x0.clone().requires_grad_(True)
            y, plink = xg, None
            for i, blk in enumerate(st.blocks):
                link = fs.SwinLink(sc[2 * i + 1])
                y = fs.SwinBlockFn.apply(y, blk, sc[2 * i], sc[2 * i + 1], plink, link, *fs.block_params(blk))
                plink = link
            (y.float()[..., :dim] * wgt).sum().backward()
            assert all(l is None or l.pre is None for l in (plink,))
            return {k: p.grad.clone() for k, p in st.named_parameters()} | {'x': xg.grad.clone()}
        finally:
            fs._LN_SCALE, fs._WGRAD_BATCH, fs._BIAS_TAIL = old
    ref = run(False, False)
    for cfg in ((True, False), (False, True), (False, False, True), (True, True, True)):
        got = run(*cfg)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (cfg, k, float((got[k].float() - ref[k].float()).abs().max()))


def test_bilinear_add_and_bn_finalize_bias_entries(hip):
    """sl_bilinear_fwd_add == clone + accumulate (same bits); sl_bn_finalize_train_bias == sl_bn_finalize_train + running_mean += momentum * bias (1e-7)."""
    from segland_amd import ops, ops_swin as osw
    torch.manual_seed(12)
    base = torch.randn(2, 24, 20, 128, device=DEV).to(torch.bfloat16)
    x = torch.randn(2, 12, 10, 128, device=DEV).to(torch.bfloat16)
    one = osw.bilinear_fwd(x, (24, 20), True, base=base)
    two = osw.bilinear_fwd(x, (24, 20), True, out=base.clone(), accumulate=True)
    assert torch.equal(one, two)
    part = torch.randn(7, 2, 128, device=DEV).abs() * 50
    part[:, 1] += 400.0
    g, b = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV)
    cb = torch.randn(96, device=DEV)
    rm0, rv0 = torch.randn(128, device=DEV), torch.rand(128, device=DEV) + 0.5
    rm1, rv1 = rm0.clone(), rv0.clone()
    a = ops.bn_finalize_train(part, 1000, g, b, rm0, rv0, 0.1, 1e-5)
    rm0[:96] += 0.1 * cb
    c = ops.bn_finalize_train(part, 1000, g, b, rm1, rv1, 0.1, 1e-5, conv_bias=cb)
    for u, v in zip(a, c):
        assert torch.equal(u, v)
    assert torch.equal(rv0, rv1) and float((rm0 - rm1).abs().max()) <= 1e-6


@pytest.mark.parametrize('inp,pl,stride,B,H', [(256, 128, 2, 4, 64), (1024, 256, 1, 4, 32), (64, 64, 1, 2, 64)])
def test_bottleneck_weight_gradient_reduces_in_one_launch_are_bit_identical(hip, inp, pl, stride, B, H):
    """The flat slab reduces of a bottleneck's 1x1 weight gradients (conv1, conv3, the downsample conv) joined in one launch (ops.WgradBatch, functional._WGRAD_BATCH):
    same summation order -> every gradient torch.equal to the one-reduce-per-layer form; a stage's first block (stride 2 / downsample branch) and identity blocks."""
    from segland_amd import functional as sf
    from segland_amd.functional import flush_num_batches_tracked
    from segland_amd.networks.backbones.resnet import Bottleneck
    torch.manual_seed(5)
    ds = None
    if stride != 1 or inp != pl * 4:
        ds = nn.Sequential(nn.Conv2d(inp, pl * 4, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(pl * 4))
    blocks = nn.Sequential(Bottleneck(inp, pl, stride=stride, dilation=1, downsample=ds), Bottleneck(pl * 4, pl, stride=1, dilation=1)).to(DEV).train()
    blocks[1].__dict__['_sl_prev'] = blocks[0]
    x = torch.randn(B, H, H, inp, device=DEV).relu_().to(torch.bfloat16)
    outs = {}
    for on in (True, False):
        old = sf._WGRAD_BATCH
        sf._WGRAD_BATCH = on
        try:
            blocks.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            y = blocks(xg)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum().backward()
            flush_num_batches_tracked()
            outs[on] = {k: p.grad.clone() for k, p in blocks.named_parameters()} | {'x': xg.grad.clone()}
        finally:
            sf._WGRAD_BATCH = old
    for k in outs[True]:
        assert torch.equal(outs[True][k], outs[False][k]), (k, float((outs[True][k].float() - outs[False][k].float()).abs().max()))


def test_bn3_apply_pass_folded_into_conv3_gradients(hip):
    """BatchNorm-backward apply pass of bn3 folded into conv3's data and weight gradient (functional._BN3_FOLD): three identity bottlenecks at a layer3-like
    shape (1024 / 256 channels, 65 536 rows, dilation 2) -- the two upper blocks receive their incoming gradient gated and reduced by the block behind (MODE 5) and take
    the folded path.  (1) it engages (bn_bwd_apply launches drop by two, the half-tile kernel serves the K = 1280 data gradient); (2) output gradient and every parameter
    gradient agree with the unfolded path to bf16 rounding of the tensor that is no longer formed (relative L2 <= 1.5e-2, cosine >= 0.9998)."""
    from segland_amd import functional as sf, ops
    from segland_amd.functional import flush_num_batches_tracked
    from segland_amd.networks.backbones.resnet import Bottleneck
    torch.manual_seed(21)
    blocks = nn.Sequential(*[Bottleneck(1024, 256, stride=1, dilation=2) for _ in range(3)]).to(DEV).train()
    with torch.no_grad():
        for m in blocks.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
    for i in (1, 2):
        blocks[i].__dict__['_sl_prev'] = blocks[i - 1]
    x = torch.randn(16, 64, 64, 1024, device=DEV).relu_().to(torch.bfloat16)      # 65 536 rows: the pixel-stationary kernel's cross-block statistics need them
    coef = torch.randn(16, 64, 64, 1024, device=DEV)
    outs, applies = {}, {}
    for on in (True, False):
        old = sf._BN3_FOLD
        sf._BN3_FOLD = on
        try:
            blocks.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            ops.PROFILER.start()
            y = blocks(xg)
            (y.float() * coef).sum().backward()
            table = ops.PROFILER.stop()
            tb = ops.PROFILER.stop_bytes()
            flush_num_batches_tracked()
            applies[on] = tb.get('bn_bwd_apply', {}).get('calls', 0)
            outs[on] = {k: p.grad.clone() for k, p in blocks.named_parameters()} | {'x': xg.grad.clone()}
        finally:
            sf._BN3_FOLD = old
    assert applies[False] - applies[True] == 2, applies
    # against the fp32 CPU oracle of the same three blocks: the folded path must be about as close to it as the unfolded one (within 25 %: it skips a bf16 rounding of dc3, but rounds the
    # folded weights; the bias is formed from the input means so that weight rounding meets centred inputs)
    from oracle import pop_oracle as po
    oracles = [po.make_bottleneck(1024, 256, 1, 2, False) for _ in range(3)]
    for b_, o in zip(blocks, oracles):
        sd = {k: v.detach().float().cpu() for k, v in b_.state_dict().items()}
        for k in list(sd):
            if k.endswith('conv1.weight') or k.endswith('conv2.weight') or k.endswith('conv3.weight'):
                sd[k] = sd[k].to(torch.bfloat16).float()
        o.load_state_dict(sd); o.train()
    xo = x.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yo = xo
    for o in oracles:
        yo = po.bottleneck_forward(o, yo)
    (yo * coef.cpu().permute(0, 3, 1, 2)).sum().backward()
    ref = {'x': xo.grad.permute(0, 2, 3, 1)}
    for i, o in enumerate(oracles):
        for k, p_ in o.named_parameters():
            ref['%d.%s' % (i, k)] = p_.grad
    worst, bad = (0.0, None), []
    for k in outs[True]:
        r = ref[k].double().reshape(-1)
        ef, eu = (float((outs[v][k].double().cpu().reshape(-1) - r).norm() / r.norm()) for v in (True, False))
        worst = max(worst, (ef / max(eu, 1e-6), k))
        print('   %-22s folded %.4f unfolded %.4f' % (k, ef, eu))
        bad = bad + [(k, ef, eu)] if ef > 1.25 * eu + 2e-3 else bad
    assert not bad, bad
    print('folded vs unfolded error against the oracle: worst ratio %.3f (%s)' % worst)


def test_gated_data_gradient_of_the_64_channel_3x3_conv_on_the_patch_kernel(hip):
    """layer1.conv2's data gradient (resnet.py:46 backward at 64 channels, 128 x 128 maps) with the ReLU gate of bn1's output and bn1's backward column sums in the store
    phase now runs on conv_c64k3_kernel (round 6; it ran on the two-stage 256 x 64 tile kernel at 79 us, the kernel's forward takes 43): the dispatch says so, the gated
    result is bit-identical to the kernel's plain data gradient with the bits applied afterwards, and the column sums match a float64 reduction of the stored tensor."""
    import ctypes
    from segland_amd import ops
    from segland_amd.ops import ConvSpec
    torch.manual_seed(31)
    B, H, W, Cn = 4, 128, 128, 64
    spec = ConvSpec(Cn, Cn, 3, 1, 1, 1)
    w = torch.randn(Cn, Cn, 3, 3, device=DEV) / 24.0
    wf = torch.empty((Cn, 3, 3, Cn), dtype=torch.bfloat16, device=DEV); wb = torch.empty((Cn, 3, 3, Cn), dtype=torch.bfloat16, device=DEV)
    ops.check(hip.sl_weight_prep(1, ops._p(w), Cn, Cn, 3, 3, ops._p(wf), ops._p(wb), ops._s()), 'weight_prep')
    dy = torch.randn(B, H, W, Cn, device=DEV).to(torch.bfloat16)
    c = torch.randn(B, H, W, Cn, device=DEV).to(torch.bfloat16)
    mean, invstd = torch.randn(Cn, device=DEV) * 0.1, torch.rand(Cn, device=DEV) + 0.5
    gate = torch.randint(0, 256, (B * H * W * Cn // 8,), dtype=torch.uint8, device=DEV)
    d = ops.conv_desc(torch.bfloat16, B, H, W, spec, None)
    assert hip.sl_conv2d_tile_config_ex(ctypes.byref(d), 1, ops.EPI_GATE) == 7016016
    r = ops.conv2d_bwd_data_bnstat(dy, wb, spec, (H, W), gate, c, mean, invstd)
    assert r is not None
    g, part = r
    assert part.shape == (B * (H // 16) * (W // 16), 2, Cn)
    plain = ops.conv2d_bwd_data(dy, wb, spec, (H, W))
    mask = ((gate.view(-1, 1).int() >> torch.arange(8, device=DEV).view(1, 8)) & 1).bool().view(B, H, W, Cn)
    want = torch.where(mask, plain, torch.zeros_like(plain))
    assert torch.equal(g, want), float((g.float() - want.float()).abs().max())
    s1 = want.double().sum((0, 1, 2))
    s2 = (want.double() * ((c.double() - mean.double()) * invstd.double())).sum((0, 1, 2))
    got = part.double().sum(0)
    assert float((got[0] - s1).abs().max() / s1.abs().max()) <= 1e-5 and float((got[1] - s2).abs().max() / s2.abs().max()) <= 1e-5
