#!/usr/bin/env python3
"""Does it learn?  Overfit one fixed synthetic batch for N steps with the drivers' optimizer (AdamW fused) in bf16 and fp32 mode;
prints the loss curves.  Guards the whole train path (weight refresh after fused steps, BN statistics, gradients) end to end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd.loss.criterion import OrthLoss
from segland_amd.networks.pspnet_pop import GFSS_Model
from segland_amd.utils.pyt_utils import get_parameters

def run(dtype, steps=60, B=4, S=256):
    torch.manual_seed(0)
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=dtype).cuda().train()
    opt = torch.optim.AdamW(get_parameters(m, lr=2e-4), lr=2e-4, weight_decay=1e-4, fused=True)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(B, 3, S, S, generator=g).cuda()
    coarse = torch.randint(0, 8, (B, S // 32, S // 32), generator=g)
    mask = coarse.repeat_interleave(32, 1).repeat_interleave(32, 2).cuda()
    out = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        d = m(img, mask)
        d['total_loss'].backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        out.append(float(d['seg_loss']))
    return out

if __name__ == '__main__':
    for dt in (torch.bfloat16, torch.float32):
        c = run(dt)
        print(str(dt), ' '.join('%.3f' % v for v in c[::6]), 'final %.3f' % c[-1])
