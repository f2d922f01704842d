#!/usr/bin/env python3
"""Time the token-map GEMMs of the Swin path (nn.Linear over NHWC tokens: forward with bias (+ residual, + GELU output), data gradient, weight gradient)
through the C ABI; kernel time from a captured HIP graph of `--reps` back-to-back launches (no launch gaps).  usage: tools/gemm_time.py --tokens 8192 --cin 384 --cout 1536 [--gelu] [--res]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops
p = argparse.ArgumentParser()
p.add_argument('--tokens', type=int, default=8192); p.add_argument('--cin', type=int, default=384); p.add_argument('--cout', type=int, default=1536)
p.add_argument('--pair-min', type=int, default=0, help='sl_debug_wgrad_pair_min'); p.add_argument('--gelu', action='store_true'); p.add_argument('--res', action='store_true'); p.add_argument('--reps', type=int, default=20)
a = p.parse_args()
dt = torch.bfloat16
if a.pair_min:
    from segland_amd import _lib
    _lib.lib().sl_debug_wgrad_pair_min(a.pair_min)
B, H, W = 8, a.tokens // 8 // 32, 32
assert B * H * W == a.tokens
spec = ops.ConvSpec(a.cin, a.cout, 1, 1, 0, 1)
x = torch.randn(B, H, W, a.cin, device='cuda').to(dt)
w = torch.randn(a.cout, a.cin, 1, 1, device='cuda') * 0.02
wf, wb = ops.weight_prep(w, dt)
bias = torch.randn(a.cout, device='cuda')
dy = torch.randn(B, H, W, a.cout, device='cuda').to(dt)
res = torch.randn(B, H, W, a.cout, device='cuda').to(dt) if a.res else None
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
gf = 2.0 * a.tokens * a.cin * a.cout / 1e9
out = []
for name, fn in [('fwd', lambda: ops.linear_fwd(x, wf, spec, bias=bias, residual=res, want_gelu=a.gelu)),
                 ('fwd_plain', lambda: ops.conv2d_fwd(x, wf, spec)),
                 ('dgrad', lambda: ops.conv2d_bwd_data(dy, wb, spec, (H, W))),
                 ('wgrad', lambda: ops.conv2d_bwd_weight(x, dy, spec))]:
    us = graph_time(fn)
    out.append('%s %6.1f us %6.0f TF/s' % (name, us, gf / us * 1e3))
print('M=%d %d->%d%s%s | ' % (a.tokens, a.cin, a.cout, ' gelu' if a.gelu else '', ' res' if a.res else '') + ' | '.join(out))
