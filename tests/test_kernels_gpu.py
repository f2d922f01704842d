"""Per-kernel parity of libsegland_hip.so (through the C ABI) against CPU fp32/fp64 references, the oracle and
the golden vectors.  Tolerances: f32 kernels 1e-4 relative to the tensor scale (fp32 MFMA is an exact fma chain, only
the summation order differs); bf16 kernels 2e-2 (inputs rounded to bf16, fp32 accumulate).  Integer outputs bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import formula as fm
from oracle import pop_oracle as po

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
DEV = 'cuda'


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2.5e-2


def nhwc(x, dtype):          # NCHW float cpu -> NHWC dtype gpu
    return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def nchw(x):                 # NHWC gpu -> NCHW float cpu
    return x.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(x, dtype):           # round a CPU float tensor through the compute dtype
    return x.to(dtype).float()


def assert_close(got, ref, dtype, what='', scale=None, factor=1.0):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    err = float((got - ref).abs().max())
    assert err <= tol(dtype) * factor * max(s, 1e-6), '%s: max abs err %g vs scale %g (%s)' % (what, err, s, dtype)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 64, 64, 1, 1, 0, 1),
    (2, 16, 16, 128, 256, 1, 1, 0, 1),
    (1, 12, 12, 64, 128, 3, 1, 1, 1),       # 144 rows: ragged row block
    (2, 16, 16, 128, 128, 3, 2, 1, 1),
    (2, 16, 16, 256, 512, 1, 2, 0, 1),
    (2, 16, 16, 64, 64, 3, 1, 2, 2),
    (2, 16, 16, 128, 64, 3, 1, 4, 4),
    (3, 1, 5, 512, 512, 1, 1, 0, 1),        # tiny M (PPM stage / head prototype rows)
    # M >= 24576 rows: the 256-row / 8-wave tile variants (N % 256, N % 128, N = 64), incl. a ragged last row block
    (8, 64, 64, 64, 256, 1, 1, 0, 1),
    (6, 64, 64, 128, 128, 3, 1, 1, 1),
    (8, 64, 64, 256, 64, 1, 1, 0, 1),
    (7, 60, 64, 64, 512, 3, 1, 2, 2),
    (2, 128, 128, 128, 128, 3, 2, 1, 1),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(hip, dtype, case):
    from segland_amd import ops
    B, H, W, Cin, Cout, k, st, pad, dil = case
    tag = 'conv%s' % (case,)
    x = rnd(fm.sym(tag + 'x', (B, Cin, H, W), 1.0), dtype)
    w = rnd(fm.sym(tag + 'w', (Cout, Cin, k, k), (3.0 / (Cin * k * k)) ** 0.5), dtype)
    x.requires_grad_(True); w.requires_grad_(True)
    y_ref = F.conv2d(x, w, None, st, pad, dil)
    gy = rnd(fm.sym(tag + 'gy', tuple(y_ref.shape), 1.0), dtype)
    y_ref.backward(gy)
    spec = ops.ConvSpec(Cin, Cout, k, st, pad, dil)
    wf, wb = ops.weight_prep(w.detach().to(DEV), dtype)
    xg = nhwc(x.detach(), dtype)
    y, part = ops.conv2d_fwd(xg, wf, spec, want_stats=True)
    assert_close(nchw(y), y_ref, dtype, 'fwd')
    # BN partial statistics = column sums / sums of squares of the fp32 result
    s = part.sum(0).cpu()
    assert_close(s[0], y_ref.detach().sum((0, 2, 3)), dtype, 'stat sum', scale=float(y_ref.abs().sum((0, 2, 3)).max()))
    assert_close(s[1], (y_ref.detach() ** 2).sum((0, 2, 3)), dtype, 'stat sq')
    gyg = nhwc(gy, dtype)
    dx = ops.conv2d_bwd_data(gyg, wb, spec, (H, W))
    assert_close(nchw(dx), x.grad, dtype, 'dgrad')
    hip.sl_debug_wgrad_tr(1)
    dw = ops.conv2d_bwd_weight(xg, gyg, spec)
    assert_close(dw, w.grad, dtype, 'wgrad')
    if dtype == torch.bfloat16:
        # the per-tap kernels' two fragment paths (the nine-tap kernel of round 5, which takes some of these shapes, has the transpose-read path only)
        hip.sl_debug_wgrad3(0)
        try:
            dw1 = ops.conv2d_bwd_weight(xg, gyg, spec)
            hip.sl_debug_wgrad_tr(0)
            dw0 = ops.conv2d_bwd_weight(xg, gyg, spec)
        finally:
            hip.sl_debug_wgrad_tr(1)
            hip.sl_debug_wgrad3(1)
        assert_close(dw0, w.grad, dtype, 'wgrad (scalar LDS path)')
        assert torch.equal(dw0, dw1), 'transpose-read and scalar fragment paths must agree bit for bit'


@pytest.mark.parametrize('B,H,W', [(4, 128, 128), (5, 120, 136)])
def test_wgrad_c64_k3_patch_kernel(hip, B, H, W):
    """layer1.conv2 (64 -> 64, 3x3) at map sizes that take the patch kernel (>= 65 536 pixels): dy tile + x patch with halo in the LDS once,
    nine taps from shifted transpose reads; full and ragged 16 x 16 tiles, against torch and against the generic (per-tap) kernel on a
    sub-batch below the threshold."""
    from segland_amd import ops
    dtype = torch.bfloat16
    x = rnd(fm.sym('c64/x%d' % H, (B, 64, H, W), 1.0), dtype)
    w = rnd(fm.sym('c64/w', (64, 64, 3, 3), (3.0 / 576) ** 0.5), dtype).requires_grad_(True)
    gy = rnd(fm.sym('c64/gy%d' % H, (B, 64, H, W), 1.0), dtype)
    F.conv2d(x, w, None, 1, 1, 1).backward(gy)
    spec = ops.ConvSpec(64, 64, 3, 1, 1, 1)
    dw = ops.conv2d_bwd_weight(nhwc(x, dtype), nhwc(gy, dtype), spec)
    assert_close(dw, w.grad, dtype, 'wgrad 64->64 3x3 (patch kernel)')
    # the same layer on two images only (32 768 pixels): generic kernel; sum of per-image-pair gradients must agree with the full batch
    if B == 4:
        parts = [ops.conv2d_bwd_weight(nhwc(x[i:i + 2], dtype), nhwc(gy[i:i + 2], dtype), spec) for i in (0, 2)]
        assert_close(parts[0] + parts[1], dw, dtype, 'patch kernel vs generic kernel')


@pytest.mark.parametrize('Cin,Cout', [(64, 256), (256, 64), (64, 64), (128, 128)])
def test_wgrad_c64_pointwise_kernel(hip, Cin, Cout):
    """1x1 layers with a 64-channel side at >= 65 536 pixels (layer1 conv1 / conv3 / downsample) take conv_wgrad_c64p_kernel: 128-pixel tiles, the whole
    [Cout][Cin] gradient in the accumulators of a block, one slab per block; against an fp32 matmul of the same bf16 operands, against the generic
    kernel on a half batch (below the threshold), into a wider gradient tensor at a channel offset, and run to run (fixed summation order)."""
    from segland_amd import ops
    dtype = torch.bfloat16
    B, H, W = 4, 128, 128
    g = torch.Generator(device='cpu').manual_seed(Cin * 3 + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(dtype).to(DEV)
    dy = torch.randn(B, H, W, Cout, generator=g).to(dtype).to(DEV)
    spec = ops.ConvSpec(Cin, Cout, 1, 1, 0, 1)
    ref = (dy.float().reshape(-1, Cout).t() @ x.float().reshape(-1, Cin)).reshape(Cout, Cin, 1, 1)
    dw = ops.conv2d_bwd_weight(x, dy, spec)
    assert_close(dw, ref, dtype, 'wgrad %d->%d 1x1 (pointwise 64-channel kernel)' % (Cin, Cout))
    assert torch.equal(dw, ops.conv2d_bwd_weight(x, dy, spec)), 'run-to-run bit stability'
    halves = [ops.conv2d_bwd_weight(x[i:i + 2].contiguous(), dy[i:i + 2].contiguous(), spec) for i in (0, 2)]
    assert_close(halves[0] + halves[1], dw, dtype, 'pointwise kernel vs generic kernel')
    wide = torch.zeros(Cout, Cin + 64, 1, 1, device=DEV)
    ops.conv2d_bwd_weight(x, dy, spec, out=wide, out_ci_off=64)
    assert torch.equal(wide[:, 64:], dw) and float(wide[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize('B,H,W', [(4, 128, 128), (5, 120, 136)])
def test_conv_c64_k3_patch_kernel(hip, B, H, W):
    """layer1.conv2 (64 -> 64, 3x3) forward and data gradient at map sizes that take the patch kernel (>= 65 536 pixels): input patch with
    halo in the LDS once, nine taps from shifted fragment reads; full and ragged 16 x 16 tiles; result, BN statistic partials (one row per
    tile) and data gradient against torch, and against the generic per-tap kernel (SEGLAND_CONV_C64K3 off is not switchable at run time, so
    the generic kernel runs on a sub-batch below the threshold)."""
    from segland_amd import ops
    dtype = torch.bfloat16
    x = rnd(fm.sym('c64f/x%d' % H, (B, 64, H, W), 1.0), dtype).requires_grad_(True)
    w = rnd(fm.sym('c64f/w', (64, 64, 3, 3), (3.0 / 576) ** 0.5), dtype)
    gy = rnd(fm.sym('c64f/gy%d' % H, (B, 64, H, W), 1.0), dtype)
    y_ref = F.conv2d(x, w, None, 1, 1, 1)
    y_ref.backward(gy)
    spec = ops.ConvSpec(64, 64, 3, 1, 1, 1)
    wf, wb = ops.weight_prep(w.to(DEV), dtype)
    xg, gyg = nhwc(x.detach(), dtype), nhwc(gy, dtype)
    y, part = ops.conv2d_fwd(xg, wf, spec, want_stats=True)
    assert part.shape[0] == B * ((H + 15) // 16) * ((W + 15) // 16)
    assert_close(nchw(y), y_ref, dtype, 'fwd 64->64 3x3 (patch kernel)')
    s = part.sum(0).cpu()
    assert_close(s[0], y_ref.detach().sum((0, 2, 3)), dtype, 'stat sum', scale=float(y_ref.abs().sum((0, 2, 3)).max()))
    assert_close(s[1], (y_ref.detach() ** 2).sum((0, 2, 3)), dtype, 'stat sq')
    dx = ops.conv2d_bwd_data(gyg, wb, spec, (H, W))
    assert_close(nchw(dx), x.grad, dtype, 'dgrad 64->64 3x3 (patch kernel)')
    # same accumulation order per output element in both kernels (tap-major, 16 channels per MFMA): the rounded results agree bit for bit
    y2 = torch.cat([ops.conv2d_fwd(xg[i:i + 2].contiguous(), wf, spec, want_stats=False)[0] for i in range(0, B - 1, 2)])
    dx2 = torch.cat([ops.conv2d_bwd_data(gyg[i:i + 2].contiguous(), wb, spec, (H, W)) for i in range(0, B - 1, 2)])
    n = y2.shape[0]
    assert_close(y2.float(), y[:n].float(), dtype, 'patch kernel vs generic kernel (fwd)')
    assert_close(dx2.float(), dx[:n].float(), dtype, 'patch kernel vs generic kernel (dgrad)')


@pytest.mark.parametrize('C_,N,dil', [(256, 256, 2), (128, 512, 4), (128, 256, 1), (256, 256, 1)])
def test_conv_3x3_patch_kernel(hip, C_, N, dil):
    """3x3 stride-1 layers (pad = dilation) on the patch kernel (conv_gemm_p9_kernel: 16 x 16-pixel output tiles, the input patch with its dilation halo in the
    LDS once per 64-channel chunk, nine taps from shifted fragment reads): forward + statistics, data gradient, data gradient + bit-gated addend must equal the
    half-tile kernel bit for bit (same K order: chunk-major, taps inside), and agree with torch on a corner crop (zero padding) and an interior crop."""
    from segland_amd import ops
    dtype = torch.bfloat16
    B, H, W = 16, 64, 64
    g = torch.Generator(device='cpu').manual_seed(C_ + N + dil)
    x = torch.randn(B, H, W, C_, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, C_, 3, 3, generator=g) * (3.0 / (9 * C_)) ** 0.5).to(dtype).float().to(DEV)
    spec = ops.ConvSpec(C_, N, 3, 1, dil, dil)
    wf, wb = ops.weight_prep(w, dtype)
    dy = torch.randn(B, H, W, N, generator=g).to(dtype).to(DEV)
    add = torch.randn(B, H, W, C_, generator=g).to(dtype).to(DEV)
    bits = torch.randint(0, 256, (add.numel() // 8,), dtype=torch.uint8, device=DEV)
    bias_v = torch.randn(N, generator=g).to(DEV); scale_v = (torch.rand(N, generator=g) + 0.5).to(DEV)
    out = {}
    try:
        for on in (0, 1):                          # half-tile kernel, patch kernel
            hip.sl_debug_conv_p9(on)
            y, part = ops.conv2d_fwd(x, wf, spec, want_stats=True)
            dx = ops.conv2d_bwd_data(dy, wb, spec, (H, W)) if N % 256 == 0 and C_ % 256 == 0 else None
            dxa = ops.conv2d_bwd_data(dy, wb, spec, (H, W), addend=add, addend_mask=bits) if dx is not None else None
            # shaped epilogues (generic store phase with the tile row map): pre-addend + statistics (factorised PPM conv), bias + ReLU, folded BN + residual + ReLU
            pre = dy
            yp, pp = ops.conv2d_fwd(x, wf, spec, pre_addend=pre, want_stats=True)
            yb, _ = ops.conv2d_fwd(x, wf, spec, bias=bias_v, relu=True)
            ya = ops.conv2d_affine_fwd(x, wf, spec, scale_v, bias_v, residual=dy, relu=True)
            out[on] = (y, part, dx, dxa, yp, pp, yb, ya)
    finally:
        hip.sl_debug_conv_p9(1)
    assert torch.equal(out[0][0], out[1][0]), 'forward: patch kernel vs half-tile kernel'
    assert_close(out[1][1].sum(0), out[0][1].sum(0), torch.float32, 'statistics', factor=10)
    if out[0][2] is not None:
        assert torch.equal(out[0][2], out[1][2]), 'data gradient'
        assert torch.equal(out[0][3], out[1][3]), 'data gradient + gated addend'
    assert torch.equal(out[0][4], out[1][4]), 'forward + pre-addend'
    assert_close(out[1][5].sum(0), out[0][5].sum(0), torch.float32, 'statistics with pre-addend', factor=10)
    assert torch.equal(out[0][6], out[1][6]), 'forward + bias + ReLU'
    assert torch.equal(out[0][7], out[1][7]), 'folded BN + residual + ReLU'
    # against torch on two crops of image 3 (top-left corner incl. zero padding; interior)
    xi = x[3].float().permute(2, 0, 1)[None]
    ref = F.conv2d(xi, w, None, 1, dil, dil)[0].permute(1, 2, 0)
    assert_close(out[1][0][3, :20, :20], ref[:20, :20], dtype, 'corner crop')
    assert_close(out[1][0][3, 24:44, 40:64], ref[24:44, 40:64], dtype, 'interior / right-edge crop')


@pytest.mark.parametrize('K,N', [(64, 256), (128, 512), (256, 1024), (256, 512), (256, 64), (128, 128), (64, 64)])
def test_conv_short_k_stationary_kernel(hip, K, N):
    """1x1 convs with 64 / 128 / 256 input channels at >= 65 536 pixels run on the pixel-stationary kernel (conv_gemm_sk_kernel): forward with
    BN statistic partials (one row per 256 pixels), plain data gradient, data gradient + addend and + bit-gated addend, against an fp32 matmul
    of the same bf16-rounded operands and bit for bit against the tile kernels (which run a
    32 768-pixel half of the same problem, below the kernel's threshold)."""
    from segland_amd import ops
    dtype = torch.bfloat16
    B, H, W = 4, 128, 128
    g = torch.Generator(device='cpu').manual_seed(K * 7 + N)
    x = torch.randn(B, H, W, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, 1, 1, generator=g) * (3.0 / K) ** 0.5).to(dtype).float().to(DEV)
    spec = ops.ConvSpec(K, N, 1, 1, 0, 1)
    wf, wb = ops.weight_prep(w, dtype)
    # forward: x [M, K] x w [N, K]^T
    y, part = ops.conv2d_fwd(x, wf, spec, want_stats=True)
    y_ref = x.float().reshape(-1, K) @ w.reshape(N, K).t()
    assert part.shape[0] == B * H * W // 256
    assert_close(y.reshape(-1, N), y_ref, dtype, 'fwd')
    yr = y.float().reshape(-1, N)
    s = part.sum(0)
    assert_close(s[0], yr.sum(0), torch.float32, 'stat sum (of the rounded result)', scale=float(yr.abs().sum(0).max()), factor=10)
    assert_close(s[1], (yr * yr).sum(0), torch.float32, 'stat sq', factor=10)
    y_half, _ = ops.conv2d_fwd(x[:2].contiguous(), wf, spec, want_stats=True)
    assert torch.equal(y_half, y[:2]), 'stationary kernel vs tile kernel (fwd) must agree bit for bit'
    # data gradient of a conv N -> K... the same GEMM shape is reached with the roles swapped: conv K_in = N_ch -> K, dy has K channels
    spec_b = ops.ConvSpec(N, K, 1, 1, 0, 1)                      # conv N -> K; its data gradient maps dy [M, K] to dx [M, N]
    wv = (torch.randn(K, N, 1, 1, generator=g) * (3.0 / K) ** 0.5).to(dtype).float().to(DEV)
    _, wvb = ops.weight_prep(wv, dtype)
    dy = torch.randn(B, H, W, K, generator=g).to(dtype).to(DEV)
    dx_ref = dy.float().reshape(-1, K) @ wv.reshape(K, N)
    dx = ops.conv2d_bwd_data(dy, wvb, spec_b, (H, W))
    assert_close(dx.reshape(-1, N), dx_ref, dtype, 'dgrad')
    add = torch.randn(B, H, W, N, generator=g).to(dtype).to(DEV)
    dx = ops.conv2d_bwd_data(dy, wvb, spec_b, (H, W), addend=add)
    assert_close(dx.reshape(-1, N), dx_ref + add.float().reshape(-1, N), dtype, 'dgrad + addend')
    gate = torch.randn(B, H, W, N, generator=g).to(dtype).to(DEV)
    one = torch.ones(N, device=DEV)
    _, bits = ops.bn_act(gate, one, torch.zeros_like(one), relu=True, want_mask=True)
    dxg = ops.conv2d_bwd_data(dy, wvb, spec_b, (H, W), addend=add, addend_mask=bits)
    assert_close(dxg.reshape(-1, N), dx_ref + (add.float() * (gate.float() > 0)).reshape(-1, N), dtype, 'dgrad + bit-gated addend')
    dxg_half = ops.conv2d_bwd_data(dy[:2].contiguous(), wvb, spec_b, (H, W), addend=add[:2].contiguous(), addend_mask=bits[:bits.numel() // 2].contiguous())
    assert torch.equal(dxg_half, dxg[:2]), 'stationary kernel vs tile kernel (dgrad + gated addend) must agree bit for bit'


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C1,C2,Cout', [(128, 64, 64), (256, 256, 256)])
def test_conv_concat_bias_relu_addend_mask(hip, dtype, C1, C2, Cout):
    from segland_amd import ops
    B, H, W = 2, 12, 12
    x1 = rnd(fm.sym('cc/x1', (B, C1, H, W), 1.0), dtype); x2 = rnd(fm.sym('cc/x2', (B, C2, H, W), 1.0), dtype)
    w = rnd(fm.sym('cc/w', (Cout, C1 + C2, 3, 3), 0.05), dtype)
    bias = fm.sym('cc/b', (Cout,), 0.5)
    x = torch.cat([x1, x2], 1).requires_grad_(True); wr = w.clone().requires_grad_(True)
    y_ref = F.relu(F.conv2d(x, wr, bias, 1, 1, 1))
    spec = ops.ConvSpec(C1 + C2, Cout, 3, 1, 1, 1)
    wf, wb = ops.weight_prep(w.to(DEV), dtype)
    a, b2 = nhwc(x1, dtype), nhwc(x2, dtype)
    y, _ = ops.conv2d_fwd(a, wf, spec, x2=b2, bias=bias.to(DEV), relu=True)
    assert_close(nchw(y), y_ref, dtype, 'concat fwd')
    gy = rnd(fm.sym('cc/gy', tuple(y_ref.shape), 1.0), dtype)
    pre = F.conv2d(x, wr, None, 1, 1, 1)
    pre.backward(gy)
    gyg = nhwc(gy, dtype)
    dw = ops.conv2d_bwd_weight(a, gyg, spec, x2=b2)
    assert_close(dw, wr.grad, dtype, 'concat wgrad')
    add = rnd(fm.sym('cc/add', (B, C1 + C2, H, W), 1.0), dtype); msk = rnd(fm.sym('cc/msk', (B, C1 + C2, H, W), 1.0), dtype)
    dx = ops.conv2d_bwd_data(gyg, wb, spec, (H, W), addend=nhwc(add, dtype), mask_src=nhwc(msk, dtype))
    assert_close(nchw(dx), (x.grad + add) * (msk > 0), dtype, 'dgrad + addend, masked')
    # addend gated by the ReLU bits of another tensor (the bottleneck shortcut gradient dout * relu'(out))
    gate = rnd(fm.sym('cc/gate', (B, C1 + C2, H, W), 1.0), dtype)
    one = torch.ones(C1 + C2, device=DEV)
    _, bits = ops.bn_act(nhwc(gate, dtype), one, torch.zeros_like(one), relu=True, want_mask=True)
    dx = ops.conv2d_bwd_data(gyg, wb, spec, (H, W), addend=nhwc(add, dtype), addend_mask=bits)
    assert_close(nchw(dx), x.grad + add * (gate > 0), dtype, 'dgrad + bit-gated addend')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C_,rows', [(64, 2 * 16 * 16), (512, 2 * 3 * 3), (2048, 70)])
def test_bn_train_eval_fwd_bwd(hip, dtype, C_, rows):
    from segland_amd import ops
    x = rnd(fm.sym('bn/x%d' % C_, (rows, C_), 2.0) + 0.3, dtype)
    res = rnd(fm.sym('bn/r%d' % C_, (rows, C_), 1.0), dtype)
    gamma = 0.8 + 0.4 * fm.uniform01('bn/g', C_).float(); beta = fm.sym('bn/b', (C_,), 0.2)
    rm, rv = fm.sym('bn/rm', (C_,), 0.1), (0.9 + 0.2 * fm.uniform01('bn/rv', C_)).float()
    gy = rnd(fm.sym('bn/gy%d' % C_, (rows, C_), 1.0), dtype)
    for train in (True, False):
        xr = x.clone().requires_grad_(True); g_ = gamma.clone().requires_grad_(True); b_ = beta.clone().requires_grad_(True)
        rm_r, rv_r = rm.clone(), rv.clone()
        pre = F.batch_norm(xr.t().reshape(1, C_, rows, 1).permute(2, 1, 0, 3), rm_r, rv_r, g_, b_, train, 0.1, 1e-5)
        pre = pre.permute(2, 1, 0, 3).reshape(C_, rows).t()
        y_ref = F.relu(pre + res)
        y_ref.backward(gy)
        xg = x.to(DEV).to(dtype); dev = lambda t: t.to(DEV)
        rm_g, rv_g = dev(rm.clone()), dev(rv.clone())
        if train:
            # statistics as the conv epilogue would deliver them: per-block (sum, sumsq)
            xf = xg.float()
            part = torch.stack([xf.sum(0), (xf * xf).sum(0)]).unsqueeze(0).contiguous()
            mean, invstd, scale, shift = ops.bn_finalize_train(part, rows, dev(gamma), dev(beta), rm_g, rv_g)
            assert_close(rm_g, rm_r, torch.float32, 'running_mean'); assert_close(rv_g, rv_r, torch.float32, 'running_var')
        else:
            mean, invstd, scale, shift = ops.bn_finalize_eval(dev(gamma), dev(beta), rm_g, rv_g)
        y = ops.bn_act(xg, scale, shift, residual=res.to(DEV).to(dtype), relu=True)
        assert_close(y, y_ref, dtype, 'bn_act train=%s' % train)
        dx, dres, dgamma, dbeta = ops.bn_bwd(gy.to(DEV).to(dtype), y, xg, mean, invstd, dev(gamma), train=train, want_dres=True)
        assert_close(dx, xr.grad, dtype, 'bn dx train=%s' % train, factor=4)
        assert_close(dres, gy * (y_ref > 0), dtype, 'bn dres')
        assert_close(dgamma, g_.grad, dtype, 'dgamma', factor=4); assert_close(dbeta, b_.grad, dtype, 'dbeta', factor=4)
        # the 1-bit ReLU mask emitted by the forward gates the backward exactly like the activation itself
        y2, bits = ops.bn_act(xg, scale, shift, residual=res.to(DEV).to(dtype), relu=True, want_mask=True)
        assert torch.equal(y2, y) and bits.dtype == torch.uint8 and bits.numel() == y.numel() * y.element_size() // 16
        dx2, _, dg2, db2 = ops.bn_bwd(gy.to(DEV).to(dtype), None, xg, mean, invstd, dev(gamma), train=train, mask=bits)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)


@pytest.mark.parametrize('dtype', DTYPES)
def test_stem(hip, dtype):
    from segland_amd import ops
    B, H, W = 2, 64, 48
    img = fm.formula_image(B, H, W, 'stem/img')
    w = fm.sym('stem/w', (64, 3, 7, 7), (6.0 / 147) ** 0.5).requires_grad_(True)
    c_ref = F.conv2d(img, w, None, 2, 3)
    c0, part = ops.stem_conv_fwd(img.to(DEV), w.detach().to(DEV), dtype, True)
    assert_close(nchw(c0), c_ref, dtype, 'stem conv')
    s = part.sum(0).cpu()
    # bf16: the MFMA kernel rounds image and weights to bf16 (like every other conv of the path), the statistics are those of ITS accumulators
    assert_close(s[0], c_ref.detach().sum((0, 2, 3)), dtype, 'stem stat sum', scale=float(c_ref.abs().sum((0, 2, 3)).max()))
    assert_close(s[1], (c_ref.detach() ** 2).sum((0, 2, 3)), dtype, 'stem stat sq')
    # BN(eval-style affine) + ReLU + maxpool and its backward, on the kernel's own (rounded) conv output
    scale = (0.8 + 0.4 * fm.uniform01('stem/sc', 64)).float(); shift = fm.sym('stem/sh', (64,), 0.3)
    cr = nchw(c0).requires_grad_(True)
    a = F.relu(cr * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    p_ref = F.max_pool2d(a, 3, 2, 1)
    gp = rnd(fm.sym('stem/gp', tuple(p_ref.shape), 1.0), dtype)
    p_ref.backward(gp)
    pooled, idx = ops.stem_bn_relu_pool(c0, scale.to(DEV), shift.to(DEV), True)
    assert_close(nchw(pooled), p_ref, dtype, 'pool fwd')
    g0 = ops.stem_pool_relu_bwd(nhwc(gp, dtype), idx, c0, scale.to(DEV), shift.to(DEV))
    # d(bn out): undo the affine part of autograd's chain to compare the masked pool gradient itself
    g_ref = cr.grad / scale.view(1, -1, 1, 1)
    assert_close(nchw(g0), g_ref, dtype, 'pool+relu bwd')
    gc = rnd(fm.sym('stem/gc', tuple(c_ref.shape), 1.0), dtype)
    c_ref.backward(gc)
    dw = ops.stem_conv_bwd_weight_im2col(img.to(DEV), nhwc(gc, dtype))
    assert_close(dw, w.grad, dtype, 'stem wgrad (im2col + MFMA)')
    dw2 = ops.stem_conv_bwd_weight_direct(img.to(DEV), nhwc(gc, dtype))          # bf16: the fused MFMA kernel; fp32: the VALU kernel
    assert_close(dw2, w.grad, dtype, 'stem wgrad (direct)')
    assert torch.equal(ops.stem_conv_bwd_weight(img.to(DEV), nhwc(gc, dtype)), dw2 if dtype == torch.bfloat16 else dw)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hw', [12, 16, 64])
def test_ppm_pool_and_upsample(hip, dtype, hw):
    from segland_amd import ops
    B, Cf, Cs, sizes = 2, 128, 64, (1, 2, 3, 6)
    x = rnd(fm.sym('ppm/x%d' % hw, (B, Cf, hw, hw), 1.0), dtype).requires_grad_(True)
    pools = [F.adaptive_avg_pool2d(x, (s, s)) for s in sizes]
    xg = nhwc(x.detach(), dtype)
    pooled = ops.ppm_pool_fwd(xg, sizes)
    off = 0
    for s, pr in zip(sizes, pools):
        got = pooled[off:off + B * s * s].float().cpu().view(B, s, s, Cf).permute(0, 3, 1, 2)
        assert pooled.dtype == torch.float32
        assert_close(got, pr, torch.float32, 'pool level %d' % s); off += B * s * s
    # backward of the pooling, fused with the direct (concat) gradient
    gps = [rnd(fm.sym('ppm/gp%d_%d' % (hw, s), (B, Cf, s, s), 1.0), dtype) for s in sizes]
    gdir = rnd(fm.sym('ppm/gd%d' % hw, (B, Cf, hw, hw), 1.0), dtype)
    (sum((p * g).sum() for p, g in zip(pools, gps)) + (x * gdir).sum()).backward()
    dpooled = torch.cat([g.permute(0, 2, 3, 1).reshape(-1, Cf) for g in gps]).contiguous().to(DEV)      # always float
    dcat = torch.zeros((B, hw, hw, 64 + Cf), dtype=dtype, device=DEV)
    dcat[..., 64:] = nhwc(gdir, dtype)
    dx = ops.ppm_pool_bwd(dpooled, (B, hw, hw, Cf), dtype, sizes, dcat=dcat, cat_off=64)
    assert_close(nchw(dx), x.grad, dtype, 'pool bwd', factor=2)
    # upsample (align_corners=False) of stage maps and its backward
    stages = [rnd(fm.sym('ppm/st%d_%d' % (hw, s), (B, Cs, s, s), 1.0), dtype).requires_grad_(True) for s in sizes]
    ups = torch.cat([F.interpolate(t, size=(hw, hw), mode='bilinear', align_corners=False) for t in stages], 1)
    stage_rows = torch.cat([t.detach().permute(0, 2, 3, 1).reshape(-1, Cs) for t in stages]).contiguous().to(DEV)  # always float
    pri = ops.ppm_upsample_fwd(stage_rows, (B, hw, hw, Cf), sizes, dtype)
    assert_close(nchw(pri), ups, dtype, 'upsample fwd')
    gu = rnd(fm.sym('ppm/gu%d' % hw, tuple(ups.shape), 1.0), dtype)
    ups.backward(gu)
    dcat2 = torch.zeros((B, hw, hw, 4 * Cs + 64), dtype=dtype, device=DEV)
    dcat2[..., :4 * Cs] = nhwc(gu, dtype)
    dst = ops.ppm_upsample_bwd(dcat2, (B, hw, hw, Cf), sizes, Cs)
    ref = torch.cat([t.grad.permute(0, 2, 3, 1).reshape(-1, Cs) for t in stages])
    assert_close(dst, ref, dtype, 'upsample bwd', factor=4)


@pytest.mark.parametrize('B,K,N', [(2, 64, 64), (3, 96, 576), (16, 2048, 512), (16, 512, 4608), (16, 4608, 512)])
def test_ppm_rows_gemm(hip, B, K, N):
    """Grouped skinny GEMM over the pyramid rows (stage convs pspnet_pop.py:12-16 + the factorised prior GEMMs): exact fp32 MFMA vs a
    float64 matmul, with and without split-K, plus the BN statistic partials."""
    from segland_amd import ops
    sizes = (1, 2, 3, 6)
    rows = ops.ppm_rows(B, sizes)
    x = fm.sym('rg/x%d_%d' % (K, N), (rows, K), 1.0)
    w = fm.sym('rg/w%d_%d' % (K, N), (len(sizes), N, K), 1.0)
    y, part = ops.ppm_rows_gemm(x.to(DEV), w.to(DEV), B, sizes, want_stats=True)
    y2, none = ops.ppm_rows_gemm(x.to(DEV), w.to(DEV), B, sizes)
    assert none is None and torch.equal(y, y2)
    off, grp = 0, ops.ppm_stat_groups(B, sizes)
    for k, s in enumerate(sizes):
        n = B * s * s
        ref = x[off:off + n].double() @ w[k].double().t()
        got = y[off:off + n].cpu().double()
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6, 'level %d' % s
        ps = part[grp[k]:grp[k + 1]].cpu().double().sum(0)
        assert float((ps[0] - ref.sum(0)).abs().max() / ref.abs().sum(0).max()) < 1e-5
        assert float((ps[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).max()) < 1e-5
        off += n


@pytest.mark.parametrize('repeat', [1, 2])
def test_adamw_multi_matches_torch(hip, repeat):
    """One-launch AdamW (csrc/optim.hip) against torch.optim.AdamW over 3 iterations, two parameter groups (lr x10 / wd 0 as in
    utils/pyt_utils.py:216-249), ragged sizes; repeat=2 = the reference's two steps per iteration (train_base.py:262-264)."""
    from segland_amd.optim import AdamW
    shapes = [(64, 3, 7, 7), (513,), (256, 64, 1, 1), (7, 512), (1,), (128, 128, 3, 3)]
    mine = [fm.sym('adam/p%d' % i, s, 1.0).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    groups = lambda ps: [dict(params=ps[:3], lr=1e-3), dict(params=ps[3:], lr=1e-2, weight_decay=0.0)]
    o1 = AdamW(groups(mine), lr=1e-3, weight_decay=1e-2)
    o2 = torch.optim.AdamW(groups(ref), lr=1e-3, weight_decay=1e-2, foreach=True)
    from segland_amd.optim import clip_coefficient
    for it in range(3):
        for i, (a, b) in enumerate(zip(mine, ref)):
            g = fm.sym('adam/g%d_%d' % (it, i), tuple(a.shape), 0.5).to(DEV)
            a.grad, b.grad = g.clone(), g.clone()
        norm, coef = clip_coefficient(mine, 5.0)                 # clipping folded into the kernel vs clip_grad_norm_ + step
        assert float(coef) < 1.0
        o1.step(repeat=repeat, grad_scale=coef)
        norm_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        assert abs(float(norm) - float(norm_ref)) <= 1e-5 * float(norm_ref)
        for _ in range(repeat):
            o2.step()
    for a, b in zip(mine, ref):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), float((a - b).abs().max())
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert set(s1['state'][0]) == set(s2['state'][0]) and float(s1['state'][0]['step']) == float(s2['state'][0]['step']) == 3 * repeat
    assert float((s1['state'][5]['exp_avg_sq'] - s2['state'][5]['exp_avg_sq']).abs().max()) < 1e-7


def test_weight_prep_batched(hip):
    """One-launch tiled weight prep == the per-conv kernel, bit for bit (bf16 and fp32 entries, 1x1 and 3x3, ragged channel counts)."""
    from segland_amd import functional as sf
    convs = [torch.nn.Conv2d(64, 128, 3, bias=False), torch.nn.Conv2d(96, 64, 1, bias=False), torch.nn.Conv2d(256, 192, 3, bias=False),
             torch.nn.Conv2d(2048, 512, 1, bias=False)]
    dts = [torch.bfloat16, torch.float32, torch.bfloat16, torch.float32]
    for k, c in enumerate(convs):
        c.weight.data = fm.sym('wp/%d' % k, tuple(c.weight.shape), 1.0)
        c.to(DEV)
    plan = sf._PrepPlan()
    plan.refresh(convs, dts)
    for c, d in zip(convs, dts):
        wf, wb = ops_mod().weight_prep(c.weight, d)
        assert torch.equal(c.weight._sl_prep[3], wf) and torch.equal(c.weight._sl_prep[4], wb)


def ops_mod():
    from segland_amd import ops
    return ops


def test_loss_golden_g3(hip):
    from segland_amd import ops
    g = golden('g3_loss')
    for preds_tag, mask_tag, K, rows, tot, dp in [('g3/preds', 'g3/mask', 8, 5, 'seg', 'dpreds'), ('g3/preds12', 'g3/mask12', 12, 3, 'seg12', 'dpreds12')]:
        preds = fm.sym(preds_tag, (2, K, 8, 8), 2.0)
        target = fm.formula_mask(2, 64, 64, K, tag=mask_tag, block=8, ignore_rows=rows)
        out = ops.upsample_ce_fwd(preds.to(DEV), target.to(DEV), 255)
        np.testing.assert_allclose(out[0].item(), g[tot], rtol=2e-5)
        assert out[1].item() == float((target != 255).sum())
        dl = ops.upsample_ce_bwd(preds.to(DEV), target.to(DEV), out, torch.ones(1, device=DEV), 255)
        np.testing.assert_allclose(dl.cpu().numpy(), g[dp], rtol=2e-4, atol=2e-7)


def test_loss_vs_oracle_512(hip):
    from segland_amd import ops
    preds = fm.sym('l512/p', (2, 8, 64, 64), 3.0).requires_grad_(True)
    target = fm.formula_mask(2, 512, 512, 8, tag='l512/m')
    up = F.interpolate(preds, size=(512, 512), mode='bilinear', align_corners=True)
    ref = F.cross_entropy(up, target, ignore_index=255)
    (ref * 0.7).backward()
    out = ops.upsample_ce_fwd(preds.detach().to(DEV), target.to(DEV), 255)
    np.testing.assert_allclose(out[0].item(), ref.item(), rtol=2e-5)
    dl = ops.upsample_ce_bwd(preds.detach().to(DEV), target.to(DEV), out, torch.full((1,), 0.7, device=DEV), 255)
    np.testing.assert_allclose(dl.cpu().numpy(), preds.grad.numpy(), rtol=1e-3, atol=1e-8)
    # all-ignored target: mean over zero valid pixels is NaN in the reference too
    t2 = torch.full((1, 64, 64), 255, dtype=torch.int64)
    out2 = ops.upsample_ce_fwd(preds.detach()[:1].contiguous().to(DEV), t2.to(DEV), 255)
    assert np.isnan(out2[0].item()) and out2[1].item() == 0


@pytest.mark.parametrize('K,hw,HW', [(8, (64, 64), (512, 512)), (12, (16, 20), (125, 160)), (3, (8, 8), (8, 8)), (8, (9, 7), (40, 33)), (16, (5, 6), (64, 64))])
def test_loss_backward_tiled_equals_gather(hip, K, hw, HW):
    """The tiled backward (each pixel's softmax once per 4x4 cell tile, separable weights in two LDS passes) against the per-cell gather kernel
    (the fallback for footprints that do not fit the LDS) and torch autograd: ignored rows, odd sizes, ragged tiles,
    the identity resize, 16 classes."""
    import subprocess, sys, os
    from segland_amd import ops
    preds = fm.sym('lt/p%d' % K, (2, K, hw[0], hw[1]), 3.0)
    target = fm.formula_mask(2, HW[0], HW[1], K, tag='lt/m%d' % K, block=8, ignore_rows=min(5, HW[0] // 2))
    pr = preds.clone().requires_grad_(True)
    ref = F.cross_entropy(F.interpolate(pr, size=HW, mode='bilinear', align_corners=True), target, ignore_index=255)
    (ref * 0.7).backward()
    gs = torch.full((1,), 0.7, device=DEV)
    out = ops.upsample_ce_fwd(preds.to(DEV), target.to(DEV), 255)
    dl = ops.upsample_ce_bwd(preds.to(DEV), target.to(DEV), out, gs, 255)
    np.testing.assert_allclose(dl.cpu().numpy(), pr.grad.numpy(), rtol=1e-3, atol=2e-8)
    assert torch.isfinite(dl).all()


def test_pseudo_label_argmax_iou_bitexact(hip):
    from segland_amd import ops
    logits = fm.sym('pl/l', (2, 5, 64, 64), 2.0)
    mask = fm.formula_mask(2, 512, 512, 8, tag='pl/m', ignore_rows=0)
    ref = mask.clone()
    for b in range(2):
        po.pseudo_label(logits[b], ref[b], 7)
    mg = mask.clone().to(DEV)
    ops.pseudo_label_(logits.to(DEV), mg, 7)
    up = F.interpolate(logits, size=(512, 512), mode='bilinear', align_corners=True)
    top2 = up.topk(2, dim=1).values
    tied = (top2[:, 0] - top2[:, 1]) < 1e-6
    diff = (mg.cpu() != ref) & ~tied
    assert int(diff.sum()) == 0
    am = ops.upsample_argmax(logits.to(DEV), (512, 512)).cpu()
    assert int(((am.long() != up.argmax(1)) & ~tied).sum()) == 0
    g = golden('g10_iou')
    pred = (fm.uniform01('g10/pred', 2 * 64 * 64) * 8).floor().long().reshape(2, 64, 64)
    tgt = (fm.uniform01('g10/tgt', 2 * 64 * 64) * 8).floor().long().reshape(2, 64, 64)
    tgt[0, :5] = 255
    h = ops.iou_hist(pred.to(torch.uint8).to(DEV), tgt.to(DEV), 8, 255).cpu().numpy()
    assert np.array_equal(h[0], g['inter'].astype(np.int64))
    assert np.array_equal(h[1] + h[2] - h[0], g['union'].astype(np.int64))
    assert np.array_equal(h[2], g['target'].astype(np.int64))


def test_masked_average_pooling_g9(hip):
    from segland_amd import ops
    feat = fm.sym('g9/feat', (2, 32, 8, 8), 1.0)
    m = (fm.uniform01('g9/mask', 2 * 64 * 64).reshape(2, 1, 64, 64) > 0.5).float()
    proto = ops.masked_avg_pool(nhwc(feat, torch.float32), m.view(2, 64, 64).contiguous().to(DEV))
    np.testing.assert_allclose(proto.cpu().numpy(), golden('g9_map')['proto'].reshape(-1), rtol=2e-5, atol=1e-6)


def test_layout_roundtrip(hip):
    from segland_amd import ops
    x = fm.sym('lay/x', (2, 24, 5, 7), 1.0)
    a = ops.nchw_f32_to_nhwc(x.to(DEV), torch.float32)
    assert torch.equal(a.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw_f32(a).cpu(), x)
