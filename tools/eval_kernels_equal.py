import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import torch
from segland_amd import ops
from segland_amd.networks.pspnet_pop import GFSS_Model
torch.manual_seed(0)
m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8).cuda().eval()
g = torch.Generator(device='cpu').manual_seed(1)
img = torch.randn(16, 3, 512, 512, generator=g).cuda()
with torch.no_grad():
    out = m(img).float().contiguous()
    am = ops.upsample_argmax(out, (512, 512))
print('logits', hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], 'argmax', hashlib.sha1(am.cpu().numpy().tobytes()).hexdigest()[:16], float(out.abs().mean()))
