#!/usr/bin/env python3
"""Weight-gradient launch time (kernel + slab reduce, HIP events) of the 3x3 layers of ResNet-50 / 101 at the bench shape: nine-tap kernel vs per-tap kernels."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops, _lib
p = argparse.ArgumentParser()
p.add_argument('--B', type=int, default=16); p.add_argument('--iters', type=int, default=20)
a = p.parse_args()
L = _lib.lib()
dt = torch.bfloat16
SHAPES = [(64, 2048, 512, 1), (64, 512, 512, 4), (64, 256, 256, 2), (64, 128, 128, 1), (128, 128, 128, 1), (32, 512, 512, 1)]
print('# B=%d; ms per launch incl. slab reduce' % a.B)
for hw, cin, cout, dil in SHAPES:
    spec = ops.ConvSpec(cin, cout, 3, 1, dil, dil)
    x = torch.randn(a.B, hw, hw, cin, device='cuda').to(dt)
    dy = torch.randn(a.B, hw, hw, cout, device='cuda').to(dt)
    gf = 2.0 * a.B * hw * hw * cin * cout * 9 / 1e9
    res = []
    for on in (0, 1, 0, 1):
        L.sl_debug_wgrad3(on)
        fn = lambda: ops.conv2d_bwd_weight(x, dy, spec)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.iters): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / a.iters)
    print('%4dx%-4d %4d->%-4d d%d   per-tap %.3f / %.3f ms (%.0f TFLOP/s)   nine-tap %.3f / %.3f ms (%.0f TFLOP/s)' %
          (hw, hw, cin, cout, dil, res[0], res[2], gf / min(res[0], res[2]), res[1], res[3], gf / min(res[1], res[3])))
L.sl_debug_wgrad3(1)
