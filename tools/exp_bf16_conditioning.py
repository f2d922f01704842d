"""CPU-only experiment behind tests/test_round2_gpu.py::test_c2_train_mode_bf16_gate: how well conditioned is the TRAIN-mode gradient of
PSPNet-POP ResNet-50 against bf16 rounding -- measured with the fp32 oracle alone (no HIP code involved).

The oracle's conv / BN helpers are wrapped so that either (a) only the stem conv output, or (b) every conv and BN output AND its gradient
is rounded to bf16 (a straightforward bf16 implementation).  The backbone / head gradients of the rounded run are compared with the
unrounded run: at He-init (random or structured data, full or damped residual branches) and along 24 AdamW steps on one structured batch.

    python tools/exp_bf16_conditioning.py            # ~10 min on 8 cores; result: profiles/r2_bf16_conditioning.txt
"""
import copy
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import formula as fm            # noqa: E402
from oracle import pop_oracle as po         # noqa: E402


class RoundBF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


MODE, CNT = [None], [0]
_cv0, _bn0 = po._cv, po._bn


def _cv(x, c):
    y = _cv0(x, c)
    CNT[0] += 1
    return RoundBF.apply(y) if (MODE[0] == 'all' or (MODE[0] == 'stem' and CNT[0] == 1)) else y


def _bn(x, b):
    y = _bn0(x, b)
    return RoundBF.apply(y) if MODE[0] == 'all' else y


po._cv, po._bn = _cv, _bn


def random_batch(B, size, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, size, size, generator=g), torch.randint(0, 8, (B, size, size), generator=g)


def structured_batch(B, size, seed):
    mask = fm.formula_mask(B, size, size, 8, 'c2/mask%d' % seed, block=32, ignore_rows=size // 10)
    g = torch.Generator().manual_seed(seed)
    color = torch.randn(8, 3, generator=g)
    idx = mask.clone(); idx[idx == 255] = 0
    return color[idx].permute(0, 3, 1, 2).contiguous() + 0.5 * torch.randn(B, 3, size, size, generator=g), mask


def grads(model, img, mask, mode):
    MODE[0], CNT[0] = mode, 0
    sd = copy.deepcopy(model.state_dict())
    model.zero_grad()
    d = model(img, mask)
    d['total_loss'].backward()
    model.load_state_dict(sd)                         # undo the running-statistics update
    bb = torch.cat([p.grad.reshape(-1) for k, p in model.named_parameters() if 'backbone' in k]).double()
    hd = torch.cat([p.grad.reshape(-1) for k, p in model.named_parameters() if 'backbone' not in k]).double()
    MODE[0] = None
    return bb, hd, float(d['seg_loss'].detach())


def cos(a, b):
    return float(a @ b / (a.norm() * b.norm()))


def make(gamma3=1.0):
    torch.manual_seed(1234)
    o = po.PopOracle(n_base=7, criterion=po.OrthLossOracle(255), backbone='resnet50')
    with torch.no_grad():
        nn.init.normal_(o.base_emb)            # off the orthogonal init: d|gram|/d(emb) is a sign of rounding noise there
        for n, m in o.named_modules():
            if n.endswith('bn3'):
                m.weight.fill_(gamma3)
    return o.train()


def main():
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    print('# rounded-oracle vs unrounded-oracle gradient, ResNet-50 PSPNet-POP, train-mode BN, batch 8 x 128x128')
    for data, fn in (('random', random_batch), ('structured', structured_batch)):
        img, mask = fn(8, 128, 5)
        for g3 in (1.0, 0.25, 0.1):
            o = make(g3)
            g0, h0, _ = grads(o, img, mask, None)
            for mode in ('stem', 'all'):
                g1, h1, _ = grads(o, img, mask, mode)
                print('He-init  data %-10s bn3.gamma %.2f  bf16 rounding %-4s : backbone cosine %.4f rel.L2 %.3f | head cosine %.4f'
                      % (data, g3, mode, cos(g0, g1), float((g0 - g1).norm() / g0.norm()), cos(h0, h1)), flush=True)
    img, mask = structured_batch(8, 128, 5)
    o = make(1.0)
    opt = torch.optim.AdamW(o.parameters(), lr=1e-3)
    for step in range(25):
        if step in (0, 4, 8, 16, 24):
            g0, h0, l0 = grads(o, img, mask, None)
            g1, h1, _ = grads(o, img, mask, 'stem')
            g2, h2, _ = grads(o, img, mask, 'all')
            print('after %2d AdamW steps (seg loss %.4f): stem rounding: backbone cosine %.4f head %.4f | all rounded: backbone cosine %.4f rel.L2 %.3f head %.4f'
                  % (step, l0, cos(g0, g1), cos(h0, h1), cos(g0, g2), float((g0 - g2).norm() / g0.norm()), cos(h0, h2)), flush=True)
        opt.zero_grad()
        d = o(img, mask)
        d['total_loss'].backward()
        torch.nn.utils.clip_grad_norm_(o.parameters(), 5.0)
        opt.step()


if __name__ == '__main__':
    main()
