import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import formula as fm, pop_oracle as po
from segland_amd import ops
from segland_amd.functional import conv_bn_fwd, conv_bn_bwd, flush_num_batches_tracked
from segland_amd.networks.backbones.resnet import Bottleneck
import torch.nn.functional as F
name = sys.argv[1] if len(sys.argv) > 1 else 's1_id'
CASES = {'s1_ds': (64, 64, 1, 1, True), 's1_id': (256, 64, 1, 1, False), 'd4_id': (1024, 256, 1, 4, False)}
inp, pl, st, dil, ds = CASES[name]
dt = torch.float32
ora = po.make_bottleneck(inp, pl, st, dil, ds)
sd = {k: fm.formula_tensor('g5' + name + '/' + k, v) for k, v in ora.state_dict().items()}
ora.load_state_dict(sd); ora.train()
x = fm.sym('g5%s/x' % name, (2, inp, 16, 16), 1.0).relu_().requires_grad_(True)
# oracle with retained intermediates
c1 = po._cv(x, ora.conv1); a1 = F.relu(po._bn(c1, ora.bn1)); c2 = po._cv(a1, ora.conv2); a2 = F.relu(po._bn(c2, ora.bn2)); c3 = po._cv(a2, ora.conv3)
pre = po._bn(c3, ora.bn3) + x; out = F.relu(pre)
for t in (c1, a1, c2, a2, c3, pre): t.retain_grad()
coef = fm.sym('g5%s/coef' % name, tuple(out.shape), 1.0)
(out * coef).sum().backward()
blk = Bottleneck(inp, pl, stride=st, dilation=dil, downsample=None)
blk.load_state_dict(sd); blk.cuda().train()
nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda().to(dt)
nc = lambda t: t.detach().float().cpu().permute(0, 3, 1, 2)
def err(a, b, what):
    a, b = nc(a), b.detach()
    print('%-10s maxerr %.3e scale %.3e  rel %.3e' % (what, (a - b).abs().max(), b.abs().max(), (a - b).abs().max() / b.abs().max()))
xg = nh(x)
gc1, ga1, m1, i1 = conv_bn_fwd(xg, blk.conv1, blk.bn1, relu=True)
gc2, ga2, m2, i2 = conv_bn_fwd(ga1, blk.conv2, blk.bn2, relu=True)
gc3, gout, m3, i3 = conv_bn_fwd(ga2, blk.conv3, blk.bn3, relu=True, residual=xg)
err(gc1, c1, 'c1'); err(ga1, a1, 'a1'); err(gc2, c2, 'c2'); err(ga2, a2, 'a2'); err(gc3, c3, 'c3'); err(gout, out, 'out')
dout = nh(coef)
da2, dw3, dg3, db3, dres = conv_bn_bwd(dout, gout, gc3, ga2, blk.conv3, blk.bn3, m3, i3, True, True, want_dres=True)
err(dres, pre.grad, 'dres'); err(da2, a2.grad, 'da2')
da1, dw2, dg2, db2, _ = conv_bn_bwd(da2, ga2, gc2, ga1, blk.conv2, blk.bn2, m2, i2, True, True)
err(da1, a1.grad, 'da1')
dxm, dw1, dg1, db1, _ = conv_bn_bwd(da1, ga1, gc1, xg, blk.conv1, blk.bn1, m1, i1, True, True, addend=None)
dx_main_ref = x.grad - pre.grad
err(dxm, dx_main_ref, 'dx_main')
dxa, _, _, _, _ = conv_bn_bwd(da1, ga1, gc1, xg, blk.conv1, blk.bn1, m1, i1, True, True, addend=dres)
err(dxa, x.grad, 'dx_total')
d = (nc(dxa) - x.grad).abs()
idx = np.unravel_index(int(d.argmax()), d.shape); print('worst at', idx, 'got', nc(dxa)[idx].item(), 'ref', x.grad[idx].item(), 'dres ref', pre.grad[idx].item(), 'main ref', dx_main_ref[idx].item())
print('num elements with err > 1e-3:', int((d > 1e-3).sum()), 'of', d.numel())
print('mask flips out:', int(((nc(gout) > 0) != (out > 0)).sum()), 'a1:', int(((nc(ga1) > 0) != (a1 > 0)).sum()), 'a2:', int(((nc(ga2) > 0) != (a2 > 0)).sum()))
# autograd path
from conftest import golden
g = golden('g5_bottleneck_' + name)
blk.zero_grad()
xg2 = nh(x).requires_grad_(True)
y = blk(xg2)
(y.float() * nh(coef)).sum().backward()
print('autograd path vs oracle:'); err(xg2.grad, x.grad, 'dx')
gd = torch.from_numpy(g['dx'])
print('oracle vs golden dx', (x.grad[:, ::4] - gd).abs().max().item(), 'gpu vs golden', (nc(xg2.grad)[:, ::4] - gd).abs().max().item())
