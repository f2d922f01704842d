#!/usr/bin/env python3
"""Race screen for the pipelined conv / wgrad kernels: every launch of a shape must be bit-identical to the first one (the LDS-DMA
pipelines are ordered only by counted waits + barriers; an early read shows up as run-to-run differences long before a tolerance
test notices).  Shapes = the layers of the benchmark network."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops

torch.manual_seed(0)
dt = torch.bfloat16
bad = 0
# background HBM traffic on a second stream perturbs load latencies (a race that hides behind a fast DMA shows up under load)
side = torch.cuda.Stream()
hog_a = torch.empty(1 << 28, dtype=torch.uint8, device='cuda'); hog_b = torch.empty_like(hog_a)
for (B, hw, cin, cout, k, dil) in [(16, 64, 512, 512, 3, 4), (16, 64, 2048, 512, 3, 1), (16, 64, 256, 1024, 1, 1), (16, 64, 1024, 256, 1, 1),
                                   (16, 64, 2048, 512, 1, 1), (16, 128, 64, 256, 1, 1), (16, 128, 64, 64, 3, 1), (7, 60, 256, 256, 3, 2)]:
    spec = ops.ConvSpec(cin, cout, k, 1, dil * (k // 2), dil)
    x = torch.randn(B, hw, hw, cin, device='cuda').to(dt)
    w = torch.randn(cout, cin, k, k, device='cuda') * 0.05
    wf, wb = ops.weight_prep(w, dt)
    dy = torch.randn(B, hw, hw, cout, device='cuda').to(dt)
    add = torch.randn(B, hw, hw, cin, device='cuda').to(dt)
    bits = torch.randint(0, 256, (add.numel() // 8,), dtype=torch.uint8, device='cuda')
    ref = None
    for it in range(int(os.environ.get('RACE_ITERS', '25'))):
        if it % 2 == 1:
            with torch.cuda.stream(side):
                for _ in range(4):
                    hog_b.copy_(hog_a)
        y, part = ops.conv2d_fwd(x, wf, spec, want_stats=True)
        dx = ops.conv2d_bwd_data(dy, wb, spec, (hw, hw), addend=add, addend_mask=bits)
        dw = ops.conv2d_bwd_weight(x, dy, spec)
        cur = (y, part, dx, dw)
        if ref is None:
            ref = tuple(t.clone() for t in cur)
        else:
            for name, a, b in zip(('fwd', 'stats', 'dgrad', 'wgrad'), ref, cur):
                if not torch.equal(a, b):
                    bad += 1
                    print('MISMATCH', (B, hw, cin, cout, k, dil), name, 'iter', it, float((a.float() - b.float()).abs().max()))
    print('shape', (B, hw, cin, cout, k, dil), 'ok' if not bad else 'so far %d mismatches' % bad, flush=True)
print('RACE SCREEN', 'CLEAN' if bad == 0 else 'FAILED (%d)' % bad)
sys.exit(1 if bad else 0)
