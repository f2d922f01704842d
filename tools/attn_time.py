#!/usr/bin/env python3
"""Window attention forward / backward at the four Swin-T stage shapes of the bench (B = 8, 512 x 512): kernel time from a captured HIP graph of back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops, ops_swin as osw
from segland_amd.ops_swin import pad_to
def graph_time(fn, reps=10):
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
for (hw, Cn, heads) in ((128, 96, 3), (64, 192, 6), (32, 384, 12), (16, 768, 24)):
    for shift in (0, 3):
        P, P3 = pad_to(Cn), pad_to(3 * Cn)
        qkv = torch.randn(8, hw, hw, P3, device='cuda').to(torch.bfloat16)
        bias = torch.randn(3 * Cn, device='cuda') * 0.5
        rel = torch.randn(heads, 49, 49, device='cuda') * 0.5
        dout = torch.randn(8, hw, hw, P, device='cuda').to(torch.bfloat16)
        f = graph_time(lambda: osw.window_attention_fwd(qkv, bias, rel, Cn, heads, shift, P))
        b = graph_time(lambda: osw.window_attention_bwd(qkv, bias, rel, dout, Cn, heads, shift, batch=ops.ColsumBatch()))      # the batch is never run: the kernel alone
        print('%3d x %3d  C %3d heads %2d shift %d: fwd %6.1f us   bwd %6.1f us' % (hw, hw, Cn, heads, shift, f, b))
