#!/usr/bin/env python3
"""Host time of the phases of a replayed bucket step at world size 1 over RCCL (ResNet-50, batch 16): how long the Python side needs to issue graph A1 / A2 / B and the
all-reduces, against the GPU time of the step.  Measured: 8.2 ms of host time per 25.8 ms step (hipGraphLaunch of the ~450-node first graph 4.9 ms, second 2.1, third 0.7, all-reduce calls 0.15):
the host runs ahead of the GPU, the 1.9 % the bucket step costs at world size 1 is on the GPU side of the eager section between the graphs."""
import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29561'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import torch, torch.distributed as dist
dist.init_process_group('nccl', rank=0, world_size=1)
from segland_amd import bucket_step
from segland_amd.loss.criterion import OrthLoss
from segland_amd.networks.pspnet_pop import GFSS_Model
from segland_amd.optim import AdamW
from segland_amd.utils.pyt_utils import get_parameters
m = GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, backbone='resnet50', dilated=True, os=8).cuda().train()
rep = bucket_step.BucketedReplica(m, cap_mb=64)
opt = AdamW(get_parameters(m, lr=1e-4), lr=1e-4, weight_decay=1e-4)
step = bucket_step.GraphedBucketStep(rep, opt, double_step=True, warmup=2)
img = torch.randn(16, 3, 512, 512, device='cuda'); mask = torch.randint(0, 8, (16, 512, 512), device='cuda')
for _ in range(6): step(img, mask)
torch.cuda.synchronize()
# host time of each phase of a replayed step
import segland_amd.bucket_step as bs
T = {}
orig_run = rep._run
def timed_run(parts, img, mask, call):
    def c2(fn, *a):
        t = time.perf_counter(); call(fn, *a); T[('call', parts.index(fn))] = T.get(('call', parts.index(fn)), 0) + time.perf_counter() - t
    orig_ar = rep.all_reduce
    def ar(which=None, async_op=False):
        t = time.perf_counter(); r = orig_ar(which, async_op); T['allreduce'] = T.get('allreduce', 0) + time.perf_counter() - t; return r
    rep.all_reduce = ar
    try: return orig_run(parts, img, mask, c2)
    finally: rep.all_reduce = orig_ar
rep._run = timed_run
t0 = time.perf_counter()
N = 40
for _ in range(N): step(img, mask)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host loop %.2f ms/step, +sync tail %.2f ms total; graphs %d' % ((t1 - t0) / N * 1e3, (t2 - t1) * 1e3, len(step.graph)))
for k, v in T.items(): print(k, '%.3f ms/step' % (v / N * 1e3))
