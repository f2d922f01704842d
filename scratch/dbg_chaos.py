import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import formula as fm
from segland_amd.networks.pspnet_pop import GFSS_Model
m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.float32)
fm.load_formula_weights(m); m = m.cuda().train()
img = fm.formula_image(2, 512, 512, 'g6/img').cuda()
def run(im, perturb=0.0):
    feats = []
    with torch.no_grad():
        x = m.backbone.forward_base_in(im)
        if perturb: x = x * (1 + perturb * torch.randn_like(x))
        feats.append(('stem', x))
        for li, stage in enumerate((m.backbone.layer1, m.backbone.layer2, m.backbone.layer3, m.backbone.layer4)):
            for bi, blk in enumerate(stage):
                x = blk(x); feats.append(('l%d.%d' % (li + 1, bi), x))
        f = m.decoder(x); feats.append(('dec', f))
        p, _, _ = m._head(f); feats.append(('preds', p))
    return feats
torch.manual_seed(0)
a = run(img); b = run(img, 2.3e-3)
for (n, x), (_, y) in zip(a, b):
    print('%-8s relL2 %.3e' % (n, (x - y).norm() / x.norm()))
