import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_round2_gpu as t
from segland_amd.optim import AdamW
from segland_amd.train_base import train_iteration
from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
img, mask = t._structured_batch(16, 512, seed=7)
img, mask = img.cuda(), mask.cuda()
torch.manual_seed(99)
m32 = t._model('resnet101', dtype=torch.float32); t._round_weights_to_bf16_(m32); m32 = m32.cuda().train()
m16 = t._model('resnet101', dtype=torch.bfloat16).cuda().train()
opt = AdamW(get_parameters(m32, lr=1e-3), lr=1e-3, weight_decay=1e-4)
sc = NativeScalerWithGradNormCount()
for step in range(41):
    if step in (8, 16, 24, 32, 40):
        sd = {k: v.detach().clone() for k, v in m32.state_dict().items()}
        m16.load_state_dict(sd)
        with torch.no_grad():
            for mm in m16.modules():
                if isinstance(mm, torch.nn.Conv2d) and mm.weight.shape[1] >= 32: mm.weight.copy_(mm.weight.to(torch.bfloat16).float())
        m32b = m32
        for m in (m32b, m16):
            m.zero_grad(set_to_none=True)
            d = m(img, mask); d['total_loss'].backward()
        print(step, float(d['seg_loss'].detach()), t._hip_group_cosines(m16, m32b), flush=True)
        m32.load_state_dict(sd)
    d, _ = train_iteration(m32, opt, sc, img, mask, double_step=False)
