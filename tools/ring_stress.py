#!/usr/bin/env python3
"""Repeatability stress of the ring GEMM kernel's round-6 schedule (a stage is waited for where its first fragments are read; the slot of stage i-1 is refilled behind a
barrier that sits in front of the last k-step): every shape runs N times on the same operands -- forward and data gradient -- and every result must be bit-identical to the
first one AND equal to a float64 reference product within bf16 rounding.  A missed wait or a slot refilled too early shows as a run-to-run difference.
usage: tools/ring_stress.py [--reps 300]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import _lib, ops
from segland_amd.ops import ConvSpec
p = argparse.ArgumentParser(); p.add_argument('--reps', type=int, default=300); a = p.parse_args()
dev = 'cuda'
L = _lib.lib()
# (B, H, W, Cin, Cout, k, dil): Swin stage 3 / 4 linears (64 x 128 and 128 x 128 tiles, 3 / 4 stages), the fine-tune pair's frozen layers, Swin stage 1 / 2 (256 x 128, 128 x 192), a 3x3
SHAPES = [(8, 32, 32, 1536, 384, 1, 1), (8, 32, 32, 384, 1536, 1, 1), (8, 32, 32, 384, 1152, 1, 1), (8, 16, 16, 3072, 768, 1, 1), (8, 16, 16, 768, 768, 1, 1),
          (2, 64, 64, 1024, 256, 1, 1), (2, 64, 64, 256, 256, 3, 2), (2, 64, 64, 2048, 512, 1, 1), (8, 128, 128, 128, 384, 1, 1), (8, 128, 128, 384, 128, 1, 1),
          (8, 64, 64, 768, 192, 1, 1), (8, 64, 64, 192, 576, 1, 1), (8, 64, 64, 128, 128, 3, 1), (3, 40, 24, 256, 128, 1, 1)]
bad = 0
for (B, H, W, Ci, Co, k, dil) in SHAPES:
    torch.manual_seed(B * 131 + Ci + Co)
    spec = ConvSpec(Ci, Co, k, 1, dil * (k // 2), dil)
    x = torch.randn(B, H, W, Ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5)
    wf = torch.empty((Co, k, k, Ci), dtype=torch.bfloat16, device=dev); wb = torch.empty((Ci, k, k, Co), dtype=torch.bfloat16, device=dev)
    ops.check(L.sl_weight_prep(_lib.SL_BF16, ops._p(w), Co, Ci, k, k, ops._p(wf), ops._p(wb), ops._s()), 'weight_prep')
    dy = torch.randn(B, H, W, Co, device=dev).to(torch.bfloat16)
    y0, _ = ops.conv2d_fwd(x, wf, spec)
    d0 = ops.conv2d_bwd_data(dy, wb, spec, (H, W))
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wf.float().permute(0, 3, 1, 2), padding=dil * (k // 2), dilation=dil).permute(0, 2, 3, 1)
    e = float((y0.float() - ref).abs().max() / ref.abs().max())
    diff = 0
    for r in range(a.reps):
        y, _ = ops.conv2d_fwd(x, wf, spec)
        d = ops.conv2d_bwd_data(dy, wb, spec, (H, W))
        if r % 25 == 24 or r == a.reps - 1:
            diff += int(not torch.equal(y, y0)) + int(not torch.equal(d, d0))
    torch.cuda.synchronize()
    print('%-34s fwd max err %.2e of scale; %d reps fwd + dgrad: %s' % (str((B, H, W, Ci, Co, k, dil)), e, a.reps, 'bit-identical' if diff == 0 else '%d DIFFERENT' % diff))
    bad += diff + int(e > 1.2e-2)
print('OK' if bad == 0 else 'FAILED (%d)' % bad)
sys.exit(1 if bad else 0)
