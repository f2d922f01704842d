import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo')
import torch, bench
from segland_amd import networks
from segland_amd.loss.criterion import OrthLoss
name = sys.argv[1]
kw = dict(dilated=True, os=8, backbone='resnet50') if name == 'pspnet_pop' else dict(backbone='swin-t')
B = 2
m = getattr(networks, name).GFSS_Model(n_base=7, criterion=OrthLoss(255), pretrained_model=None, compute_dtype=torch.bfloat16, **kw).cuda().train()
opt = bench.make_optimizer(m); params = [p for p in m.parameters() if p.requires_grad]
img, mask = bench.synthetic_batch(B, 256, 'cuda')
for _ in range(4): bench.train_step(m, opt, img, mask, params, True)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): bench.train_step(m, opt, img, mask, params, True)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
