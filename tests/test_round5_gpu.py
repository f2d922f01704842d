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
