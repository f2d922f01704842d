#!/usr/bin/env python3
"""Where the host spends its time between two replays of the captured training step (does hipGraphLaunch return before the graph has run?).
usage: python tools/graph_host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from segland_amd import graph_step, networks
from segland_amd.loss.criterion import OrthLoss
m = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, dilated=True, os=8, backbone='resnet50').cuda().train()
opt = bench.make_optimizer(m)
params = [p for p in m.parameters() if p.requires_grad]
batches = [bench.synthetic_batch(16, 512, 'cuda', seed=k) for k in range(4)]
fn = lambda m_, o_, s_, im_, mk_, double_step=True: (bench.train_step(m_, o_, im_, mk_, params, double_step, 1), None)
g = graph_step.GraphedTrainStep(fn, m, opt, None, double_step=True, warmup=2)
for k in range(6):
    g(*batches[k % 4])
torch.cuda.synchronize()
T = {'copy': 0.0, 'prepare': 0.0, 'replay': 0.0}
n = 20
t_all = time.perf_counter()
for k in range(n):
    img, mask = batches[k % 4]
    t0 = time.perf_counter()
    for dst, src in zip(g.static_in, (img, mask)):
        dst.copy_(src, non_blocking=True)
    t1 = time.perf_counter()
    opt.graph_prepare()
    t2 = time.perf_counter()
    g.graph.replay()
    t3 = time.perf_counter()
    T['copy'] += t1 - t0; T['prepare'] += t2 - t1; T['replay'] += t3 - t2
t_issue = time.perf_counter() - t_all
torch.cuda.synchronize()
t_total = time.perf_counter() - t_all
print('per step: host copy %.3f ms, graph_prepare %.3f ms, replay() call %.3f ms; host issue loop %.3f ms/step, wall incl. final sync %.3f ms/step'
      % tuple(1e3 * x / n for x in (T['copy'], T['prepare'], T['replay'], t_issue, t_total)))
