"""Tensor-level wrappers over the Swin-POP part of the C ABI (include/segland_hip.h, "Swin-POP path").

Token maps are NHWC tensors [B,H,W,P] in the compute dtype whose first C channels are real and whose pad (P - C channels, zero) exists only
where C is not a multiple of 64 (Swin-T/S stage 1 and the 96-wide decoder: P = 128).  nn.Linear layers run on the MFMA conv kernels as 1x1
convolutions over these maps (ops.conv2d_fwd / conv2d_bwd_data / conv2d_bwd_weight)."""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import SlResizeDesc, SlWinDesc, check
from .ops import _f32, _p, _s, dt


def pad_to(c, m=128):
    """Channel pitch of a C-channel map: C itself when it is a multiple of 64, else the next multiple of 128."""
    return c if c % 64 == 0 else (c + m - 1) // m * m


# --------------------------------------------------------------------------------------------- patch embedding
def patch_embed_fwd(img, w, bias, dtype, pitch):
    B, _, H, W = img.shape
    Cn = w.shape[0]
    out = torch.empty((B, (H + 3) // 4, (W + 3) // 4, pitch), dtype=dtype, device=img.device)
    check(_lib.lib().sl_patch_embed_fwd(dt(dtype), _p(img), _p(w), _p(bias), _p(out), B, H, W, Cn, pitch, _s()), 'patch_embed_fwd')
    return out


def patch_embed_bwd(img, dy, Cn):
    """(dw [C,3,4,4], dbias [C]) float: im2col of the 4x4 patches + the MFMA weight-gradient GEMM (the same route as the ResNet stem)."""
    B, _, H, W = img.shape
    ntok = B * ((H + 3) // 4) * ((W + 3) // 4)
    P = dy.shape[-1]
    col = torch.empty((1, 1, ntok, 64), dtype=dy.dtype, device=img.device)
    check(_lib.lib().sl_patch_im2col(dt(dy), _p(img), _p(col), B, H, W, _s()), 'patch_im2col')
    dwp = ops.conv2d_bwd_weight(col, dy.view(1, 1, ntok, P), ops.ConvSpec(64, P, 1))
    return dwp[:Cn, :48, 0, 0].reshape(Cn, 3, 4, 4), ops.colsum_rows(dy)[:Cn].contiguous()


# --------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, Cn, out_pitch=None, eps=1e-5, want_stats=True):
    px = x.shape[-1]
    py = px if out_pitch is None else out_pitch
    rows = x.numel() // px
    y = torch.empty(x.shape[:-1] + (py,), dtype=x.dtype, device=x.device)
    stats = _f32((rows, 2), x.device) if want_stats else None
    check(_lib.lib().sl_layernorm_fwd(dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(stats), rows, Cn, px, py, eps, _s()), 'layernorm_fwd')
    return y, stats


def layernorm_bwd(dy, x, gamma, stats, Cn, addend=None, want_param_grads=True, dx_pitch=None, batch=None, row_scale=None):
    """(dx, dgamma, dbeta); dx gets `addend` added (the gradient that bypasses the norm through the residual connection).
    row_scale [B] float: dx is returned as the pair (dx, dx * row_scale[b]) -- the second is what scale_add(dx, row_scale) would make, from the same launch."""
    pdy, px = dy.shape[-1], x.shape[-1]
    pdx = px if dx_pitch is None else dx_pitch
    rows = x.numel() // px
    dx = torch.empty(x.shape[:-1] + (pdx,), dtype=x.dtype, device=x.device)
    L = _lib.lib()
    part = _f32((L.sl_layernorm_bwd_rows(dt(x), rows, Cn, pdx), 2, Cn), x.device) if want_param_grads else None
    if row_scale is not None:
        dx2 = torch.empty_like(dx)
        check(L.sl_layernorm_bwd_scaled(dt(x), _p(dy), _p(x), _p(gamma), _p(stats), _p(addend), _p(dx), _p(dx2), _p(row_scale), rows // x.shape[0], _p(part), rows, Cn,
                                        pdy, px, pdx, _s()), 'layernorm_bwd_scaled')
        dx = (dx, dx2)
    else:
        check(L.sl_layernorm_bwd(dt(x), _p(dy), _p(x), _p(gamma), _p(stats), _p(addend), _p(dx), _p(part), rows, Cn, pdy, px, pdx, _s()), 'layernorm_bwd')
    if not want_param_grads:
        return dx, None, None
    tot = batch.add(part) if batch is not None else ops.colsum(part)       # batch: filled by batch.run()
    return dx, tot[0], tot[1]


# --------------------------------------------------------------------------------------------- elementwise
def gelu_fwd(h):
    y = torch.empty_like(h)
    check(_lib.lib().sl_gelu_fwd(dt(h), _p(h), _p(y), h.numel(), _s()), 'gelu_fwd')
    return y


def gelu_bwd(h, dy):
    dh = torch.empty_like(h)
    check(_lib.lib().sl_gelu_bwd(dt(h), _p(h), _p(dy), _p(dh), h.numel(), _s()), 'gelu_bwd')
    return dh


def scale_add(x, scale, addend=None, per_channel=False, Cn=None):
    """addend + x * scale[b]  (per_channel: x * scale[b][c], channel pad forced to zero)."""
    B, P = x.shape[0], x.shape[-1]
    out = torch.empty_like(x)
    check(_lib.lib().sl_scale_add(dt(x), _p(x), _p(scale), _p(addend), _p(out), B, x.numel() // (B * P), P if Cn is None else Cn, P, int(per_channel), _s()), 'scale_add')
    return out


def merge_gather(x, Cn):
    B, H, W, P = x.shape
    xm = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * Cn), dtype=x.dtype, device=x.device)
    check(_lib.lib().sl_patch_merge_gather(dt(x), _p(x), _p(xm), B, H, W, Cn, P, _s()), 'patch_merge_gather')
    return xm


def merge_scatter(dxm, shape, Cn):
    B, H, W, P = shape
    dx = torch.empty(shape, dtype=dxm.dtype, device=dxm.device)
    check(_lib.lib().sl_patch_merge_scatter(dt(dxm), _p(dxm), _p(dx), B, H, W, Cn, P, _s()), 'patch_merge_scatter')
    return dx


# --------------------------------------------------------------------------------------------- bilinear resize
def _rd(dtype, B, h, w, H, W, Cn, sp, so, dp, do, align, acc, src_f32=False):
    return SlResizeDesc(dt(dtype), B, h, w, H, W, Cn, sp, so, dp, do, int(align), int(acc), int(src_f32))


def bilinear_fwd(x, size, align, out=None, out_off=0, Cn=None, src_off=0, accumulate=False, base=None):
    """F.interpolate(x, size, mode='bilinear', align_corners=align) on an NHWC map (channel window [src_off, src_off+Cn) -> [out_off, ...)).
    base: out = base + interpolate(x) (same shape as out; no copy of base first)."""
    B, h, w, P = x.shape
    Cn = P if Cn is None else Cn
    if out is None:
        out = torch.empty((B, size[0], size[1], Cn), dtype=x.dtype, device=x.device)
    d = _rd(out.dtype, B, h, w, size[0], size[1], Cn, P, src_off, out.shape[-1], out_off, align, accumulate, src_f32=(x.dtype == torch.float32))
    if base is not None:
        assert not accumulate and base.shape == out.shape and base.dtype == out.dtype and base.is_contiguous()
        check(_lib.lib().sl_bilinear_fwd_add(C.byref(d), _p(x), _p(base), _p(out), _s()), 'bilinear_fwd_add')
        return out
    check(_lib.lib().sl_bilinear_fwd(C.byref(d), _p(x), _p(out), _s()), 'bilinear_fwd')
    return out


def bilinear_bwd(dy, src_hw, align, out=None, out_off=0, Cn=None, dy_off=0, accumulate=False):
    """Gradient wrt the SOURCE of bilinear_fwd: dy [B,H,W,Pd] -> [B,h,w,.]."""
    B, H, W, Pd = dy.shape
    Cn = Pd if Cn is None else Cn
    if out is None:
        out = torch.empty((B, src_hw[0], src_hw[1], Cn), dtype=dy.dtype, device=dy.device)
    d = _rd(dy.dtype, B, src_hw[0], src_hw[1], H, W, Cn, out.shape[-1], out_off, Pd, dy_off, align, accumulate, src_f32=(out.dtype == torch.float32))
    check(_lib.lib().sl_bilinear_bwd(C.byref(d), _p(dy), _p(out), _s()), 'bilinear_bwd')
    return out


# --------------------------------------------------------------------------------------------- window attention
def win_desc(dtype, B, H, W, Cn, heads, qkv_pitch, out_pitch, shift):
    return SlWinDesc(dt(dtype), B, H, W, Cn, heads, qkv_pitch, out_pitch, shift)


def window_attention_fwd(qkv, qkv_bias, rel_bias, Cn, heads, shift, out_pitch):
    B, H, W, P3 = qkv.shape
    out = torch.empty((B, H, W, out_pitch), dtype=qkv.dtype, device=qkv.device)
    d = win_desc(qkv.dtype, B, H, W, Cn, heads, P3, out_pitch, shift)
    check(_lib.lib().sl_window_attention_fwd(C.byref(d), _p(qkv), _p(qkv_bias), _p(rel_bias), _p(out), _s()), 'window_attention_fwd')
    return out


def relpos_table_grad(dbias, pairs, rows, qkv_bias=None):
    """d relative_position_bias_table [rows, heads] from d bias [heads, 49*49] through the constant pair lists `pairs` [rows, m] (int32, -1 padded).
    qkv_bias = (dbq_colsum [3*C], dpad [heads, 3, 32] contiguous): also returns d qkv.bias [3*C] = dbq_colsum + dpad in the bias' (3, heads, 32) order, same launch."""
    heads = dbias.shape[0]
    out = _f32((rows, heads), dbias.device)
    if qkv_bias is not None:
        dbq_in, dpad = qkv_bias
        assert dbq_in.is_contiguous() and dpad.is_contiguous() and dbq_in.numel() == 3 * heads * 32 == dpad.numel() and dbq_in.dtype == dpad.dtype == torch.float32
        dbq = torch.empty_like(dbq_in)
        check(_lib.lib().sl_relpos_table_grad_bias(_p(dbias), _p(pairs), rows, pairs.shape[1], heads, dbias[0].numel(), _p(out), _p(dbq_in), _p(dpad), _p(dbq), _s()),
              'relpos_table_grad_bias')
        return out, dbq
    check(_lib.lib().sl_relpos_table_grad(_p(dbias), _p(pairs), rows, pairs.shape[1], heads, dbias[0].numel(), _p(out), _s()), 'relpos_table_grad')
    return out


def window_attention_bwd(qkv, qkv_bias, rel_bias, dout, Cn, heads, shift, batch=None):
    """(dqkv, d rel_bias [heads,49,49], d qkv_bias through the pad tokens [3C] as a [3, heads, 32] view); batch: the two column sums are
    filled by batch.run()."""
    B, H, W, P3 = qkv.shape
    d = win_desc(qkv.dtype, B, H, W, Cn, heads, P3, dout.shape[-1], shift)
    L = _lib.lib()
    nchunk, nwin = L.sl_window_attention_bwd_chunks(C.byref(d)), L.sl_window_attention_windows(C.byref(d))
    dqkv = torch.empty_like(qkv)
    drel = _f32((nchunk, heads * 49 * 49), qkv.device)
    pad = _f32((nwin, heads * 96), qkv.device)
    check(L.sl_window_attention_bwd(C.byref(d), _p(qkv), _p(qkv_bias), _p(rel_bias), _p(dout), _p(dqkv), _p(drel), _p(pad), _s()), 'window_attention_bwd')
    if batch is not None:
        return dqkv, batch.add(drel).view(heads, 49, 49), batch.add(pad).view(heads, 3, 32).permute(1, 0, 2)
    drel_b = ops.colsum(drel).view(heads, 49, 49)
    dpad = ops.colsum(pad).view(heads, 3, 32).permute(1, 0, 2).reshape(3 * Cn)       # [q | k | v] x heads x 32, the qkv channel order
    return dqkv, drel_b, dpad
