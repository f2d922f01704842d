#!/usr/bin/env python3
"""Which Python lines launch the small torch-side kernels of a training step (torch.profiler with stacks): counts of aten ops per call site
in segland_amd/, for the launch diet of the Swin path.  usage: op_sites.py [pspnet_pop|swin_pop]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from segland_amd import networks
from segland_amd.loss.criterion import OrthLoss
model_name = sys.argv[1] if len(sys.argv) > 1 else 'swin_pop'
kw = dict(dilated=True, os=8, backbone='resnet50') if model_name == 'pspnet_pop' else dict(backbone='swin-t')
B = 16 if model_name == 'pspnet_pop' else 8
m = getattr(networks, model_name).GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, **kw).cuda().train()
opt = bench.make_optimizer(m)
params = [p for p in m.parameters() if p.requires_grad]
img, mask = bench.synthetic_batch(B, 512, 'cuda')
for _ in range(3):
    bench.train_step(m, opt, img, mask, params, True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step(m, opt, img, mask, params, True)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    site = next((s for s in (ev.stack or []) if 'segland_amd' in s or 'bench.py' in s), '?')
    sites[(ev.name, ('segland_amd/' + site.split('segland_amd/')[-1]) if 'segland_amd/' in site else site[-80:])] += len(ev.kernels)
for (name, site), n in sites.most_common(60):
    print('%4d  %-28s %s' % (n, name, site))
