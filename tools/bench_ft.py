#!/usr/bin/env python3
"""Config 4 of BASELINE.json: ft_pop.py novel-class update (1 novel + 1 base 512x512 tile per step, frozen backbone/decoder,
trainable novel prototypes + classifier_n), pairs/s on one MI355X.  Not the headline bench (bench.py)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import graph_step
from segland_amd.ft_pop import ft_graph_body, ft_iteration, ft_iteration_graphed
from segland_amd.loss.criterion import OrthLoss
from segland_amd import networks
from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters

p = argparse.ArgumentParser(); p.add_argument('--steps', type=int, default=30); p.add_argument('--warmup', type=int, default=5)
p.add_argument('--dtype', default='f32', choices=['f32', 'bf16']); p.add_argument('--pairs', type=int, default=1)
p.add_argument('--no-step-graph', action='store_true', help='kernel-by-kernel steps (default: forward + backward + clip replayed as one HIP graph, like ft_pop.py)')
p.add_argument('--model', default='pspnet_pop', choices=['pspnet_pop', 'swin_pop']); p.add_argument('--backbone', default=None)
p.add_argument('--torch-sgd', action='store_true', help='torch.optim.SGD stepped behind the graph (round 3) instead of segland_amd.optim.SGD inside it')
a = p.parse_args()
dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
torch.manual_seed(0)
kw = dict(dilated=True, os=8) if a.model == 'pspnet_pop' else {}
m = getattr(networks, a.model).GFSS_Model(n_base=7, criterion=OrthLoss(255), is_ft=True, n_novel=4, backbone=a.backbone or ('resnet50' if a.model == 'pspnet_pop' else 'swin-t'),
                                          pretrained_model=None, compute_dtype=dt, **kw).cuda()
m.init_cls_n()
from segland_amd.optim import SGD
opt = (torch.optim.SGD if a.torch_sgd else SGD)(get_parameters(m, lr=1e-3, freeze_backbone=True), lr=1e-3, momentum=0.9, weight_decay=5e-4)
B = a.pairs
img, img_b = torch.randn(B, 3, 512, 512, device='cuda'), torch.randn(B, 3, 512, 512, device='cuda')
mask = torch.randint(8, 12, (B, 512, 512), device='cuda'); mask[:, :40] = 255
mask_b0 = torch.randint(0, 8, (B, 512, 512), device='cuda')
sc = NativeScalerWithGradNormCount()
m.train_mode()
in_graph = None if a.torch_sgd else opt
graphed = None if a.no_step_graph or not graph_step.eligible(m, opt, 'cuda', need_adamw=False) else graph_step.GraphedStep(ft_graph_body(m, optimizer=in_graph), m, in_graph)
def step():
    if graphed is not None:
        return ft_iteration_graphed(graphed, opt, (img, mask, img_b, mask_b0), 'cuda')      # the graph copies mask_b into its static input
    return ft_iteration(m, opt, sc, (img, mask, img_b, mask_b0.clone()), 'cuda')
for _ in range(a.warmup): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
print(json.dumps({'metric': 'ft_pop pairs/sec (1 novel + 1 base 512x512 tile per pair)', 'value': round(B * a.steps / dt_s, 2), 'unit': 'pairs/s',
                  'ms_per_step': round(1e3 * dt_s / a.steps, 3), 'dtype': a.dtype, 'pairs_per_step': B,
                  'step_issue': ('HIP graph replay + torch SGD behind it' if a.torch_sgd else 'one HIP graph replay (SGD step inside)') if graphed is not None and graphed.graph is not None else 'kernel by kernel'}))
