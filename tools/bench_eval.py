#!/usr/bin/env python3
"""Inference throughput (eval_base.py loop body: forward + fused upsample/argmax), tiles/s at a given batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops
from segland_amd.networks.pspnet_pop import GFSS_Model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8).cuda().eval()
img = torch.randn(B, 3, 512, 512, device='cuda')
with torch.no_grad():
    for _ in range(5):
        ops.upsample_argmax(m(img).float().contiguous(), (512, 512))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        ops.upsample_argmax(m(img).float().contiguous(), (512, 512))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('eval batch %d: %.1f tiles/s (%.2f ms per forward)' % (B, 50 * B / dt, 1e3 * dt / 50))
