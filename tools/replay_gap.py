#!/usr/bin/env python3
"""What one replayed training step costs beyond its kernels: the ResNet-50 bench step as (a) GraphedTrainStep.__call__ (input copies + AdamW.graph_prepare + replay: what
bench.py and train_base time), (b) graph_prepare + replay, (c) bare replay(), 50 steps each, wall clock over the region (host queues ahead; one synchronise at the end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from segland_amd import graph_step, networks
from segland_amd.loss.criterion import OrthLoss
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, compute_dtype=torch.bfloat16, dilated=True, os=8).to(dev).train()
opt = bench.make_optimizer(m)
params = [p for p in m.parameters() if p.requires_grad]
batches = [bench.synthetic_batch(16, 512, dev, seed=k) for k in range(4)]
for k in range(3):
    bench.train_step(m, opt, *batches[k], params, True)
fn = lambda m_, o_, s_, im_, mk_, double_step=True: (bench.train_step(m_, o_, im_, mk_, params, double_step), None)
g = graph_step.GraphedTrainStep(fn, m, opt, None, double_step=True, warmup=0)
for k in range(3):
    g(*batches[k % 4])
assert g.graph is not None


def timed(step, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, step in (('GraphedTrainStep.__call__ (copies + graph_prepare + replay)', lambda k: g(*batches[k % 4])),
                   ('graph_prepare + replay', lambda k: (opt.graph_prepare(), g.graph.replay())),
                   ('bare replay', lambda k: g.graph.replay()),
                   ('GraphedTrainStep.__call__ again', lambda k: g(*batches[k % 4]))):
    print('%-62s %.3f ms per step' % (name, timed(step)))
