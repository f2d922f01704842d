#!/usr/bin/env python3
"""Effective HBM rate of the three BatchNorm streaming kernels per layer shape (bf16): rows x C as in ResNet-50 at batch 16, 512x512."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops

def timeit(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

for rows, C in [(262144, 64), (262144, 256), (65536, 128), (65536, 512), (65536, 256), (65536, 1024), (65536, 2048)]:
    x = torch.randn(rows, C, device='cuda').bfloat16(); dy = torch.randn_like(x); res = torch.randn_like(x)
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda'); mean = torch.randn(C, device='cuda'); inv = torch.rand(C, device='cuda') + 0.5
    y, mask = ops.bn_act(x, sc, sh, relu=True, want_mask=True)
    n = rows * C
    t_f = timeit(lambda: ops.bn_act(x, sc, sh, relu=True, want_mask=True))
    t_fr = timeit(lambda: ops.bn_act(x, sc, sh, residual=res, relu=True, want_mask=True))
    t_b = timeit(lambda: ops.bn_bwd(dy, None, x, mean, inv, sc, mask=mask))
    print('rows %7d C %4d | fwd %6.1f us %5.2f TB/s | fwd+res %6.1f us %5.2f TB/s | bwd (reduce+finalize+apply) %6.1f us %5.2f TB/s' %
          (rows, C, t_f * 1e3, n * 4.125 / t_f / 1e9, t_fr * 1e3, n * 6.125 / t_fr / 1e9, t_b * 1e3, n * 10.25 / t_b / 1e9))
