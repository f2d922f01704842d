#!/usr/bin/env python3
"""Time one conv shape (fwd / dgrad(+addend) / wgrad) through the C ABI with HIP events."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops
p = argparse.ArgumentParser()
p.add_argument('--B', type=int, default=16); p.add_argument('--hw', type=int, default=64)
p.add_argument('--cin', type=int, default=256); p.add_argument('--cout', type=int, default=1024)
p.add_argument('--k', type=int, default=1); p.add_argument('--dil', type=int, default=1); p.add_argument('--iters', type=int, default=20)
p.add_argument('--zeros', action='store_true', help='all-zero operands: the same instruction stream at lower switching power (DVFS check, MI355X_MICROARCH.md)')
a = p.parse_args()
dt = torch.bfloat16
spec = ops.ConvSpec(a.cin, a.cout, a.k, 1, a.dil * (a.k // 2), a.dil)
x = torch.randn(a.B, a.hw, a.hw, a.cin, device='cuda').to(dt)
w = torch.randn(a.cout, a.cin, a.k, a.k, device='cuda') * 0.02
wf, wb = ops.weight_prep(w, dt)
dy = torch.randn(a.B, a.hw, a.hw, a.cout, device='cuda').to(dt)
add = torch.randn(a.B, a.hw, a.hw, a.cin, device='cuda').to(dt)
bits = torch.randint(0, 256, (add.numel() // 8,), dtype=torch.uint8, device='cuda')
if a.zeros:
    for t in (x, dy, add, wf, wb): t.zero_()
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
M = a.B * a.hw * a.hw
gf = 2.0 * M * a.cin * a.cout * a.k * a.k / 1e9
es = 2
for name, fn, byts in [('fwd', lambda: ops.conv2d_fwd(x, wf, spec, want_stats=True), M * (a.cin + a.cout) * es),
                       ('dgrad', lambda: ops.conv2d_bwd_data(dy, wb, spec, (a.hw, a.hw)), M * (a.cin + a.cout) * es),
                       ('dgrad+add', lambda: ops.conv2d_bwd_data(dy, wb, spec, (a.hw, a.hw), addend=add), M * (2 * a.cin + a.cout) * es),
                       ('dgrad+add+bits', lambda: ops.conv2d_bwd_data(dy, wb, spec, (a.hw, a.hw), addend=add, addend_mask=bits), M * (2 * a.cin + a.cout) * es),
                       ('wgrad', lambda: ops.conv2d_bwd_weight(x, dy, spec), M * (a.cin + a.cout) * es)]:
    ms = timeit(fn)
    print('%-10s %8.3f ms  %8.1f TFLOP/s  %6.2f TB/s (activation bytes only)' % (name, ms, gf / ms, byts / ms / 1e9))
