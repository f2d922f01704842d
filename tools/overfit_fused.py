#!/usr/bin/env python3
"""Does the bench-shape step still learn with the round-3 fusions on (BN-backward statistics in the data-gradient epilogues incl. across blocks, dual BN backward, fused
prototype kernel)?  Overfits one synthetic batch (R50, bf16, 16 tiles of 512 x 512, the train_base loop body with the graph-replayed step) with the fusions on and off
and prints both loss curves: they must fall together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from segland_amd import functional as sf, graph_step, networks
from segland_amd.loss.criterion import OrthLoss


def run(fused, steps=40):
    sf._BN_FUSE = sf._BN_DUAL = sf._BN_CROSS = fused
    networks.pspnet_pop._PROTO_FUSED = fused
    torch.manual_seed(0)
    m = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, dilated=True, os=8, backbone='resnet50').cuda().train()
    opt = bench.make_optimizer(m, lr=2e-4)
    params = [p for p in m.parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(1)
    coarse = torch.randint(0, 8, (16, 16, 16), generator=g)
    mask = coarse.repeat_interleave(32, 1).repeat_interleave(32, 2).cuda()
    color = torch.randn(8, 3, generator=g)
    img = (color[mask.cpu()].permute(0, 3, 1, 2) + 0.5 * torch.randn(16, 3, 512, 512, generator=g)).contiguous().cuda()
    fn = lambda m_, o_, s_, im_, mk_, double_step=True: (bench.train_step(m_, o_, im_, mk_, params, double_step, 1), None)
    step = graph_step.GraphedTrainStep(fn, m, opt, None, double_step=True, warmup=2)
    out = []
    for i in range(steps):
        d, _ = step(img, mask)
        out.append(float(d['seg_loss']))
    return out


if __name__ == '__main__':
    a, b = run(True), run(False)
    print('fused  :', ' '.join('%.3f' % v for v in a[::4]), '... last %.4f' % a[-1])
    print('unfused:', ' '.join('%.3f' % v for v in b[::4]), '... last %.4f' % b[-1])
    assert a[-1] < 0.5 * a[0] and b[-1] < 0.5 * b[0], 'the batch is not being fitted'
    assert abs(a[-1] - b[-1]) <= 0.25 * max(a[-1], b[-1]) + 0.05, 'fused and unfused runs diverge'
    print('ok')
