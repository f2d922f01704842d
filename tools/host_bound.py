#!/usr/bin/env python3
"""Is the bench step host-bound?  Time to ENQUEUE K steps (python returns) vs time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from segland_amd.loss.criterion import OrthLoss
from segland_amd.networks.pspnet_pop import GFSS_Model

torch.manual_seed(0)
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8).to(dev).train()
opt = bench.make_optimizer(model, torch_optimizer=len(sys.argv) > 2)
params = [p for p in model.parameters() if p.requires_grad]
img, mask = bench.synthetic_batch(B, 512, dev)
for _ in range(5):
    bench.train_step(model, opt, img, mask, params, True)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    bench.train_step(model, opt, img, mask, params, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('batch %d: enqueue %.2f ms/step, complete %.2f ms/step, GPU tail after the last enqueue %.2f ms' % (B, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K, 1e3 * (t2 - t1)))
