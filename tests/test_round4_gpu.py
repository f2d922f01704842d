"""Round-4 gates: the advisor's findings of round 3 (two live forward passes through the cross-block BatchNorm fusion, the shape check of the fused prototype
kernels), small-M conv dispatch (the fine-tune pair's 8 192-row layers on split-K), the faster BatchNorm finalize kernels, the TIFF tile decode."""
import os

import numpy as np
import pytest
import torch

from oracle import formula as fm

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------------------------- advisor (round 3, medium): two forwards before the first backward
def test_two_live_forward_passes_keep_their_own_bn3_handover(hip):
    """functional.BottleneckFn hands ReLU bits / c3 of block i to block i + 1's backward and the column sums back (resnet.py:71-78 across a block boundary).  Round 3 kept
    that in module state read at BACKWARD time: loss1 = model(x1); loss2 = model(x2); loss1.backward() gated pass 1's gradient with pass 2's bits, silently.  The records
    are per forward pass now (functional._BlockLink in ctx): gradients of two interleaved passes equal the gradients of each pass run alone, bit for bit."""
    from segland_amd import ops
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    torch.manual_seed(5)
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.bfloat16).to(DEV).train()
    for mod in m.modules():                              # running statistics do not enter a train-mode gradient, but keep the passes independent of their order anyway
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0
    xs = [fm.formula_image(16, 256, 256, 'live/img%d' % i).to(DEV) for i in range(2)]
    ys = [fm.formula_mask(16, 256, 256, 8, 'live/mask%d' % i, block=16, ignore_rows=8 + 8 * i).to(DEV) for i in range(2)]
    calls = [0]
    real = ops.conv2d_bwd_data_addend_bnstat

    def counted(*a, **k):
        out = real(*a, **k)
        calls[0] += out is not None
        return out

    def grads():
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    try:
        ops.conv2d_bwd_data_addend_bnstat = counted
        alone = []
        for i in range(2):
            m.zero_grad(set_to_none=True)
            m(xs[i], ys[i])['total_loss'].backward()
            alone.append(grads())
        assert calls[0] >= 2, 'the cross-block route was not taken at this shape (%d calls)' % calls[0]
        for order in ((0, 1), (1, 0)):
            losses = [m(xs[i], ys[i])['total_loss'] for i in range(2)]          # both graphs alive
            for i in order:
                m.zero_grad(set_to_none=True)
                losses[i].backward()
                g = grads()
                bad = [k for k in alone[i] if not torch.equal(g[k], alone[i][k])]
                assert not bad, 'pass %d (backward order %s): %d gradients differ from the pass run alone, e.g. %s' % (i, order, len(bad), bad[:3])
    finally:
        ops.conv2d_bwd_data_addend_bnstat = real
