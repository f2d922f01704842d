"""Closed-form deterministic tensors for parity tests (TEST INFRASTRUCTURE).

Weights and inputs are regenerated bit-identically on every machine from
``(tensor name, flat index)`` through a 32-bit integer hash, so no weight file is ever
committed and the reference (in the build container) and the HIP path (on the GPU box)
see exactly the same numbers.  Follows the plan of SURVEY.md section 8c.
"""
import math
import zlib

import torch

_M32 = 0xFFFFFFFF


def _fmix32(x):
    # murmur3 finaliser on int64 tensors holding uint32 values
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    x = x ^ (x >> 16)
    return x


def uniform01(name, numel, salt=0):
    """float64 tensor of ``numel`` values in [0, 1), a pure function of (name, salt, index)."""
    seed = (zlib.crc32(name.encode()) + 0x9E3779B1 * (salt + 1)) & _M32
    idx = torch.arange(numel, dtype=torch.int64)
    x = (idx * 0x9E3779B1 + seed) & _M32
    x = _fmix32(x)
    x = _fmix32((x + 0x7F4A7C15) & _M32)
    return x.to(torch.float64) / 4294967296.0


def sym(name, shape, bound, salt=0):
    n = int(math.prod(shape))
    return ((uniform01(name, n, salt) * 2.0 - 1.0) * bound).to(torch.float32).reshape(shape)


def formula_tensor(name, ref):
    """Value for state_dict entry ``name`` shaped/dtyped like ``ref``."""
    shape = tuple(ref.shape)
    if name.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=ref.dtype)
    if name.endswith('running_mean'):
        return sym(name, shape, 0.1)
    if name.endswith('running_var'):
        return (0.9 + 0.2 * uniform01(name, int(math.prod(shape)))).to(torch.float32).reshape(shape)
    if name in ('base_emb', 'novel_emb'):
        return sym(name, shape, 1.0)
    if name.endswith('relative_position_index'):            # integer buffer of the Swin attention (a function of the window size)
        return ref.detach().clone()
    if name.endswith('relative_position_bias_table'):
        return sym(name, shape, 0.5)
    if len(shape) == 2:  # nn.Linear weight [out, in]: unit-gain uniform
        return sym(name, shape, math.sqrt(3.0 / shape[1]))
    if len(shape) == 4:  # conv weight, Kaiming-uniform for ReLU nets
        fan_in = shape[1] * shape[2] * shape[3]
        return sym(name, shape, math.sqrt(6.0 / fan_in))
    if len(shape) == 1 and name.endswith('weight'):  # BN gamma
        return (0.8 + 0.4 * uniform01(name, shape[0])).to(torch.float32)
    if len(shape) == 1 and name.endswith('bias'):  # BN beta / conv bias
        return sym(name, shape, 0.2)
    raise ValueError('no formula for %s %s' % (name, shape))


def formula_state_dict(model):
    return {k: formula_tensor(k, v) for k, v in model.state_dict().items()}


def load_formula_weights(model):
    sd = formula_state_dict(model)
    missing = model.load_state_dict(sd, strict=True)
    return model


def formula_image(B, H, W, tag='img'):
    """[B,3,H,W] fp32 in [-1,1]: smooth low-frequency field + hash noise (OEM tiles are (x/255-0.5)/0.5)."""
    ys = torch.arange(H, dtype=torch.float64).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    bs = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    cs = torch.arange(3, dtype=torch.float64).view(1, 3, 1, 1)
    smooth = 0.5 * torch.sin(0.031 * xs * (1 + cs) + 0.7 * bs) * torch.cos(0.023 * ys * (1 + 0.5 * cs) - 0.3 * bs)
    noise = (uniform01(tag, B * 3 * H * W).reshape(B, 3, H, W) * 2.0 - 1.0) * 0.5
    return (smooth + noise).clamp(-1.0, 1.0).to(torch.float32)


def formula_mask(B, H, W, n_class, tag='mask', block=32, ignore_rows=50, lo=0):
    """[B,H,W] int64 labels in [lo, lo+n_class): blocky regions with 10% hash noise;
    rows [0, ignore_rows) of sample 0 are 255 (ignore_index)."""
    hb, wb = (H + block - 1) // block, (W + block - 1) // block
    coarse = (uniform01(tag + '/coarse', B * hb * wb) * n_class).floor().to(torch.int64).reshape(B, hb, wb)
    m = coarse.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :H, :W].clone()
    u = uniform01(tag + '/flip', B * H * W).reshape(B, H, W)
    r = (uniform01(tag + '/val', B * H * W).reshape(B, H, W) * n_class).floor().to(torch.int64)
    m = torch.where(u < 0.1, r, m) + lo
    if ignore_rows > 0:
        m[0, :ignore_rows] = 255
    return m.contiguous()
