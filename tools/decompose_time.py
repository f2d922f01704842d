#!/usr/bin/env python3
"""Time pop_decompose forward / backward (csrc/pop_head.hip; pspnet_pop.py:107-121 orthogonal_decompose) at the row counts of the bench configurations: kernel time from a
captured HIP graph of back-to-back launches, against the bytes a pass has to move (fwd: read feats, write bg; bwd: read dg, feats, write dq)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from segland_amd import ops
dt = torch.bfloat16
def graph_time(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3
for R, Kt, what in ((8192, 11, 'fine-tune pair'), (8192, 7, 'one pair, base only'), (65536, 7, 'ResNet-50 step'), (131072, 7, 'Swin-T step (8 x 128 x 128)')):
    C = 512
    feats = torch.randn(R, C, device='cuda').to(dt)
    S = F.normalize(torch.randn(Kt, C, device='cuda'), dim=-1)
    bg = torch.empty_like(feats)
    proj = ops.pop_decompose_into(feats, S, bg)
    dg = torch.randn_like(feats); dproj = torch.randn(R, Kt, device='cuda')
    tf = graph_time(lambda: ops.pop_decompose_into(feats, S, bg))
    tb = graph_time(lambda: ops.pop_decompose_bwd(dg, feats, S, proj, dproj))
    nb = feats.numel() * 2
    print('%-28s R %6d Kt %2d | fwd %6.1f us %5.2f TB/s | bwd (+ dS partials + finalize) %6.1f us %5.2f TB/s' % (what, R, Kt, tf, 2 * nb / tf / 1e6, tb, 3 * nb / tb / 1e6))
