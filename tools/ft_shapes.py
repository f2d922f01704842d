#!/usr/bin/env python3
"""Per-layer time of the frozen feature extractor of ONE fine-tune pair (BASELINE config 4: 2 tiles of 512 x 512 -> 8 192 pixel rows from layer2 on): every distinct
conv + folded-BN shape of ResNet-50 (os 8) + the pyramid conv, graph-timed (no launch gaps), with the default dispatch and with the split-K workspace withheld.
usage: tools/ft_shapes.py [--B 2]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import _lib, ops
p = argparse.ArgumentParser(); p.add_argument('--reps', type=int, default=20); p.add_argument('--B', type=int, default=2)
a = p.parse_args()
L = _lib.lib()
dt = torch.bfloat16
def graph_time(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(a.reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * a.reps) * 1e3
# (count per forward, H, Cin, Cout, k, stride, dil, residual)
B = a.B
SHAPES = [(1, 128, 64, 64, 1, 1, 1, False), (3, 128, 64, 64, 3, 1, 1, False), (3, 128, 64, 256, 1, 1, 1, True), (2, 128, 256, 64, 1, 1, 1, False),
          (1, 128, 256, 128, 1, 1, 1, False), (1, 128, 128, 128, 3, 2, 1, False), (1, 128, 256, 512, 1, 2, 1, False),
          (4, 64, 128, 512, 1, 1, 1, True), (3, 64, 512, 128, 1, 1, 1, False), (3, 64, 128, 128, 3, 1, 1, False),
          (1, 64, 512, 256, 1, 1, 1, False), (5, 64, 1024, 256, 1, 1, 1, False), (6, 64, 256, 256, 3, 1, 2, False), (6, 64, 256, 1024, 1, 1, 1, True), (1, 64, 512, 1024, 1, 1, 1, False),
          (1, 64, 1024, 512, 1, 1, 1, False), (2, 64, 2048, 512, 1, 1, 1, False), (3, 64, 512, 512, 3, 1, 4, False), (3, 64, 512, 2048, 1, 1, 1, True), (1, 64, 1024, 2048, 1, 1, 1, False),
          (1, 64, 2048, 512, 3, 1, 1, False), (1, 64, 512, 512, 1, 1, 1, False)]
tot = {}
print('# B=%d; us per launch, TFLOP/s; split = parts of the K range (0: not split)' % B)
for cnt, H, cin, cout, k, st, dil, res in SHAPES:
    spec = ops.ConvSpec(cin, cout, k, st, dil * (k // 2), dil)
    x = torch.randn(B, H, H, cin, device='cuda').to(dt)
    w = torch.randn(cout, cin, k, k, device='cuda') * 0.02
    wf, _ = ops.weight_prep(w, dt)
    sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda') * 0.1
    Ho = spec.out_hw(H, H)[0]
    r = torch.randn(B, Ho, Ho, cout, device='cuda').to(dt) if res else None
    d = ops.conv_desc(dt, B, H, H, spec)
    need = L.sl_conv2d_affine_fwd_workspace(C.byref(d))
    parts = need // (B * Ho * Ho * cout * 4)
    y = torch.empty(B, Ho, Ho, cout, device='cuda', dtype=dt)
    us = graph_time(lambda: ops.conv2d_affine_fwd(x, wf, spec, sc, sh, residual=r, relu=True, out=y))
    us0 = graph_time(lambda: _lib.check(L.sl_conv2d_affine_fwd_ex(C.byref(d), ops._p(x), None, ops._p(wf), None, ops._p(sc), ops._p(sh), ops._p(r), 1, ops._p(y), None, 0, ops._s()), 'x')) if parts else us
    gf = 2.0 * B * Ho * Ho * cin * cout * k * k / 1e9
    cfg = L.sl_conv2d_tile_config(C.byref(d), 0)
    print('%d x  %4d -> %4d k%d s%d d%d @%3d  cfg %8d split %d : %7.1f us %6.0f TF/s   (unsplit %7.1f us)' % (cnt, cin, cout, k, st, dil, H, cfg, parts, us, gf / us * 1e3, us0))
    tot['now'] = tot.get('now', 0) + cnt * us; tot['unsplit'] = tot.get('unsplit', 0) + cnt * us0; tot['gf'] = tot.get('gf', 0) + cnt * gf
print('# sum over the forward: %.1f us (%.0f TFLOP/s), without split-K %.1f us' % (tot['now'], tot['gf'] / tot['now'] * 1e3, tot['unsplit']))
