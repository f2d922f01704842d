import sys, torch
sys.path.insert(0, '/root/repo')
from segland_amd import optim, networks
from segland_amd.loss.criterion import OrthLoss
m = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, compute_dtype=torch.bfloat16, dilated=True, os=8).cuda()
ps = [p for p in m.parameters() if p.requires_grad]
for p in ps: p.grad = torch.randn_like(p)
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for flag in (True, False, True, False):
    optim._NORM_KERNEL = flag
    r = optim.clip_coefficient(ps, 5.0)
    print('kernel' if flag else 'torch ', '%.1f us' % t(lambda: optim.clip_coefficient(ps, 5.0)), float(r[0]), float(r[1]))
