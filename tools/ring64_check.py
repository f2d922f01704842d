#!/usr/bin/env python3
"""64 x 128 ring tiles (hook sl_debug_ring64_max_tiles) against the 128 x 128 ones on inference convs of the fine-tune pair's shapes: bit-equality and time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segland_amd import _lib, ops
L = _lib.lib()
dev = 'cuda'
torch.manual_seed(0)
shapes = [(1024, 256, 1, 1), (256, 256, 3, 2), (512, 256, 1, 1), (512, 128, 1, 1), (128, 128, 3, 1), (256, 128, 1, 1), (2048, 512, 1, 1), (512, 512, 3, 4), (256, 1024, 1, 1)]
for cin, cout, k, dil in shapes:
    B, H, W = 2, 64, 64
    x = torch.randn(B, H, W, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5)
    wf, _ = ops.weight_prep(w, torch.bfloat16)
    scale, shift = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    spec = ops.ConvSpec(cin, cout, k, 1, dil if k == 3 else 0, dil)
    res = torch.randn(B, H, W, cout, device=dev).to(torch.bfloat16)
    outs, times = [], []
    for tiles in (0, 1 << 30):
        L.sl_debug_ring64_max_tiles(tiles)
        y = ops.conv2d_affine_fwd(x, wf, spec, scale, shift, relu=True, residual=res)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                y2 = ops.conv2d_affine_fwd(x, wf, spec, scale, shift, relu=True, residual=res)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 100 * 1e6)
        outs.append(y.clone())
    L.sl_debug_ring64_max_tiles(0)
    print('%4d -> %4d k%d d%d: 128x128 %6.1f us, 64x128 %6.1f us, bit-equal %s' % (cin, cout, k, dil, times[0], times[1], bool(torch.equal(outs[0], outs[1]))), flush=True)
