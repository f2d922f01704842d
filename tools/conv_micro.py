#!/usr/bin/env python3
"""Micro-driver: runs ONE conv shape (default: the PPM bottleneck 3x3 4096->512 of PSPNet-POP at batch 16) fwd / dgrad /
wgrad a few times through the C ABI.  Used under rocprofv3 (--kernel-trace / --pmc) to read per-kernel counters."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops

p = argparse.ArgumentParser()
p.add_argument('--B', type=int, default=16); p.add_argument('--hw', type=int, default=64)
p.add_argument('--cin', type=int, default=4096); p.add_argument('--cout', type=int, default=512)
p.add_argument('--k', type=int, default=3); p.add_argument('--dil', type=int, default=1)
p.add_argument('--c1', type=int, default=2048, help='channels of the first source (virtual concat); 0 = single source')
p.add_argument('--iters', type=int, default=5); p.add_argument('--what', default='fwd,dgrad,wgrad')
a = p.parse_args()
dt = torch.bfloat16
pad = a.dil * (a.k // 2)
spec = ops.ConvSpec(a.cin, a.cout, a.k, 1, pad, a.dil)
torch.manual_seed(0)
c1 = a.c1 if a.c1 else a.cin
x1 = torch.randn(a.B, a.hw, a.hw, c1, device='cuda').to(dt)
x2 = torch.randn(a.B, a.hw, a.hw, a.cin - c1, device='cuda').to(dt) if c1 < a.cin else None
w = torch.randn(a.cout, a.cin, a.k, a.k, device='cuda') * 0.02
wf, wb = ops.weight_prep(w, dt)
dy = torch.randn(a.B, a.hw, a.hw, a.cout, device='cuda').to(dt)
for it in range(a.iters):
    if 'fwd' in a.what:
        y, part = ops.conv2d_fwd(x1, wf, spec, x2=x2, want_stats=True)
    if 'dgrad' in a.what:
        dx = ops.conv2d_bwd_data(dy, wb, spec, (a.hw, a.hw), C1=(c1 if x2 is not None else None))
    if 'wgrad' in a.what:
        dw = ops.conv2d_bwd_weight(x1, dy, spec, x2=x2)
torch.cuda.synchronize()
print('done')
