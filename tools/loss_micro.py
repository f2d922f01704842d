#!/usr/bin/env python3
"""Micro-driver: the upsample + cross-entropy forward / backward of one bench step (B = 16, 8 logit channels, 64 x 64 -> 512 x 512) a few times, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops
torch.manual_seed(0)
lg = torch.randn(16, 8, 64, 64, device='cuda')
tg = torch.randint(0, 8, (16, 512, 512), device='cuda')
tg[:, :8] = 255
one = torch.ones(1, device='cuda')
for _ in range(5):
    lc = ops.upsample_ce_fwd(lg, tg, 255)
    dl = ops.upsample_ce_bwd(lg, tg, lc, one, 255)
torch.cuda.synchronize()
print('done', float(lc[0]), float(dl.abs().sum()))
