import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import formula as fm
from segland_amd.networks.pspnet_pop import GFSS_Model
def build(dt):
    m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=dt)
    fm.load_formula_weights(m); return m.cuda()
img = fm.formula_image(2, 512, 512, 'g6/img').cuda()
for mode in ('eval', 'train'):
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        m = build(dt); m.train(mode == 'train')
        feats = []
        with torch.no_grad():
            x = m.backbone.forward_base_in(img); feats.append(('stem', x))
            for li, stage in enumerate((m.backbone.layer1, m.backbone.layer2, m.backbone.layer3, m.backbone.layer4)):
                for bi, blk in enumerate(stage):
                    x = blk(x); feats.append(('l%d.%d' % (li + 1, bi), x))
            f = m.decoder(x); feats.append(('dec', f))
            p, _, _ = m._head(f); feats.append(('preds', p))
        outs[dt] = feats
    print('==', mode)
    for (n, a), (_, b) in zip(outs[torch.float32], outs[torch.bfloat16]):
        a, b = a.float(), b.float()
        print('%-8s scale %.3e  relL2 %.3e  maxrel %.3e' % (n, a.abs().max(), (a - b).norm() / a.norm(), (a - b).abs().max() / a.abs().max()))
