#!/usr/bin/env python3
"""Times of the pixel-stationary kernel's forms at the layer3 shapes (B 16, 64 x 64): conv3 forward 256 -> 1024 with statistics (MODE 1), conv1's data gradient
1024 <- 256 plain (MODE 1), + shortcut addend (MODE 2), + addend + gate + bn3 column sums (MODE 5: sl_conv2d_bwd_data_addend_bnstat).  HIP events, 20 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops
dt = torch.bfloat16
B, H = 16, 64
M = B * H * H
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(B, H, H, 256, device='cuda').to(dt)
w3 = torch.randn(1024, 256, 1, 1, device='cuda') * 0.05
wf3, _ = ops.weight_prep(w3, dt)
spec3 = ops.ConvSpec(256, 1024, 1)
ms = timeit(lambda: ops.conv2d_fwd(x, wf3, spec3, want_stats=True))
print('forward 256 -> 1024 + statistics      %7.1f us   %5.2f TB/s' % (1e3 * ms, M * (256 + 1024) * 2 / ms / 1e9))
w1 = torch.randn(256, 1024, 1, 1, device='cuda') * 0.03
_, wb1 = ops.weight_prep(w1, dt)
spec1 = ops.ConvSpec(1024, 256, 1)
dy = torch.randn(B, H, H, 256, device='cuda').to(dt)
add = torch.randn(B, H, H, 1024, device='cuda').to(dt)
c3 = torch.randn(B, H, H, 1024, device='cuda').to(dt)
gate = torch.randint(0, 256, (M * 1024 // 8,), dtype=torch.uint8, device='cuda')
mean, invstd = torch.randn(1024, device='cuda') * 0.1, torch.rand(1024, device='cuda') + 0.5
ms = timeit(lambda: ops.conv2d_bwd_data(dy, wb1, spec1, (H, H)))
print('data gradient 1024 <- 256 plain        %7.1f us   %5.2f TB/s' % (1e3 * ms, M * (256 + 1024) * 2 / ms / 1e9))
ms = timeit(lambda: ops.conv2d_bwd_data(dy, wb1, spec1, (H, H), addend=add))
print('  + shortcut addend                    %7.1f us   %5.2f TB/s' % (1e3 * ms, M * (256 + 2048) * 2 / ms / 1e9))
ms = timeit(lambda: ops.conv2d_bwd_data_addend_bnstat(dy, wb1, spec1, (H, H), add, gate, c3, mean, invstd))
print('  + addend + gate + bn3 column sums    %7.1f us   %5.2f TB/s' % (1e3 * ms, (M * (256 + 3072) * 2 + M * 128) / ms / 1e9))
