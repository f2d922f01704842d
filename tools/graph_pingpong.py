#!/usr/bin/env python3
"""Is there a per-replay cost of launching the SAME graph exec back to back?  Two captures of the same training step (same model / optimizer, own static inputs) replayed
alternately against one capture replayed every step.  Measured: no difference (26.23-26.41 vs 26.28-26.33 ms ResNet-50, 11.21-11.23 vs 11.22-11.23 ms Swin-T): relaunching the
same exec costs nothing, the ~0.5 ms gap the kernel trace shows at the top of a replayed step is not a relaunch bubble.  usage: tools/graph_pingpong.py [swin]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import graph_step
from segland_amd.loss.criterion import OrthLoss
from segland_amd.optim import AdamW
from segland_amd.train_base import train_iteration
from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
swin = len(sys.argv) > 1 and sys.argv[1] == 'swin'
if swin:
    from segland_amd.networks.swin_pop import GFSS_Model
    m, B = GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, backbone='swin-t'), 8
else:
    from segland_amd.networks.pspnet_pop import GFSS_Model
    m, B = GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, backbone='resnet50', dilated=True, os=8), 16
m = m.cuda().train()
opt = AdamW(get_parameters(m, lr=1e-4), lr=1e-4, weight_decay=1e-4)
sc = NativeScalerWithGradNormCount()
img = torch.randn(B, 3, 512, 512, device='cuda'); mask = torch.randint(0, 8, (B, 512, 512), device='cuda')
body = lambda i, k: train_iteration(m, opt, sc, i, k, double_step=True)
def flat(o): return o
s1 = graph_step.GraphedStep(lambda i, k: body(i, k)[0]['total_loss'], m, opt, warmup=2)
for _ in range(4): s1(img, mask)
cap0 = opt.capture_begin
opt.capture_begin = lambda: None                       # the second capture reuses the optimizer's capture buffers (same hyper-parameter row, same tables)
s2 = graph_step.GraphedStep(lambda i, k: body(i, k)[0]['total_loss'], m, opt, warmup=0)
s2(img, mask)
opt.capture_begin = cap0
assert s1.graph is not None and s2.graph is not None
def run(seq, n=60):
    for k in range(6): seq[k % len(seq)](img, mask)
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(n): seq[k % len(seq)](img, mask)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rep in range(2):
    print('one exec every step   %.3f ms/step' % run([s1]))
    print('two execs alternating %.3f ms/step' % run([s1, s2]))
