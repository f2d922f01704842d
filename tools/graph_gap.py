#!/usr/bin/env python3
"""Where the idle time at the top of a replayed step comes from (kernel traces show 0.2-0.45 ms between the last eager kernel in front of a graph launch and the graph's
first node).  Host-side stopwatch around the pieces of one replayed step -- the input copies, torch.cuda.CUDAGraph.replay() -- for the ResNet-50 training step and, with
`ft`, the fine-tune pair step: does replay() return at once (the host runs ahead and the gap is the runtime's) or does it block until the previous replay has finished
(the host is in lock-step and everything it does between two replays is exposed)?  usage: tools/graph_gap.py [ft]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import graph_step, networks
from segland_amd.loss.criterion import OrthLoss
from segland_amd.optim import AdamW
from segland_amd.train_base import train_iteration
from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
ft = len(sys.argv) > 1 and sys.argv[1] == 'ft'
dt = torch.bfloat16
torch.manual_seed(0)
if ft:
    from segland_amd.ft_pop import ft_graph_body
    m = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), is_ft=True, n_novel=4, backbone='resnet50', pretrained_model=None, compute_dtype=dt, dilated=True, os=8).cuda()
    m.init_cls_n(); m.train_mode()
    opt = torch.optim.SGD(get_parameters(m, lr=1e-3, freeze_backbone=True), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    args = (torch.randn(1, 3, 512, 512, device='cuda'), torch.randint(8, 12, (1, 512, 512), device='cuda'), torch.randn(1, 3, 512, 512, device='cuda'), torch.randint(0, 8, (1, 512, 512), device='cuda'))
    g = graph_step.GraphedStep(ft_graph_body(m), m)
    after = lambda: (opt.step(), opt.zero_grad())
else:
    m = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, compute_dtype=dt, dilated=True, os=8).cuda().train()
    opt = AdamW(get_parameters(m, lr=1e-4), lr=1e-4, weight_decay=1e-4)
    sc = NativeScalerWithGradNormCount()
    args = (torch.randn(16, 3, 512, 512, device='cuda'), torch.randint(0, 8, (16, 512, 512), device='cuda'))
    g = graph_step.GraphedStep(lambda i, k: train_iteration(m, opt, sc, i, k, double_step=True)[0]['total_loss'], m, opt, warmup=2)
    after = lambda: None
for _ in range(8):
    g(*args); after()
assert g.graph is not None
torch.cuda.synchronize()
N = 40
# (1) the product's call path, host stopwatch per piece
tc = tr = ta = 0.0
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    for dst, src in zip(g.static_in, args):
        dst.copy_(src, non_blocking=True)
    b = time.perf_counter()
    if g.optimizer is not None:
        g.optimizer.graph_prepare()
    g.graph.replay()
    c = time.perf_counter()
    after()
    d = time.perf_counter()
    tc += b - a; tr += c - b; ta += d - c
host_loop = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print('%s step: %.3f ms per step wall; host per step: input copies %.3f ms, graph_prepare + replay() %.3f ms, eager tail (optimizer) %.3f ms; the host loop finished %.1f ms before the GPU'
      % ('ft pair' if ft else 'ResNet-50', total / N * 1e3, tc / N * 1e3, tr / N * 1e3, ta / N * 1e3, (total - host_loop) * 1e3))
# (2) replays only, nothing eager in between
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    g.graph.replay()
h = time.perf_counter() - t0
torch.cuda.synchronize()
print('   replay() alone, back to back: %.3f ms per step wall, %.3f ms of host time per call' % ((time.perf_counter() - t0) / N * 1e3, h / N * 1e3))
# (3) with the product's bookkeeping (GraphedStep.__call__)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    g(*args); after()
h = time.perf_counter() - t0
torch.cuda.synchronize()
print('   GraphedStep.__call__ + tail: %.3f ms per step wall, %.3f ms of host time per step' % ((time.perf_counter() - t0) / N * 1e3, h / N * 1e3))
