#!/usr/bin/env python3
"""Time LayerNorm forward / backward (csrc/swin.hip) on the token maps of the four Swin-T stages (8 tiles of 512 x 512): kernel time from a captured HIP graph of back-to-back
launches, and the achieved HBM rate against the bytes each pass has to move (fwd: read x, write y; bwd: read dy, x, addend, write dx)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops_swin as osw
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
for tokens, Cn, P in ((131072, 96, 128), (32768, 192, 192), (8192, 384, 384), (2048, 768, 768)):
    x = torch.randn(8, tokens // 8 // 32, 32, P, device='cuda').to(dt)
    gamma, beta = torch.rand(Cn, device='cuda') + 0.5, torch.randn(Cn, device='cuda')
    y, st = osw.layernorm_fwd(x, gamma, beta, Cn)
    dy = torch.randn_like(x); add = torch.randn_like(x)
    tf = graph_time(lambda: osw.layernorm_fwd(x, gamma, beta, Cn))
    tb = graph_time(lambda: osw.layernorm_bwd(dy, x, gamma, st, Cn, addend=add))
    tb0 = graph_time(lambda: osw.layernorm_bwd(dy, x, gamma, st, Cn, addend=add, want_param_grads=False))
    nb = x.numel() * 2
    print('tokens %6d C %3d pitch %3d | fwd %6.1f us %5.2f TB/s | bwd (+ dgamma / dbeta partials + their finalize) %6.1f us %5.2f TB/s | bwd dx only %6.1f us %5.2f TB/s'
          % (tokens, Cn, P, tf, 2 * nb / tf / 1e6, tb, 4 * nb / tb / 1e6, tb0, 4 * nb / tb0 / 1e6))
