#!/usr/bin/env python3
"""Phase trace of window_attention_bwd_mfma2_kernel (hook sl_debug_attn_trace: s_memtime of thread 0 at the phase boundaries of each block's first window) on the four
Swin-T stages: median shader ticks per phase over the blocks (the counters of different XCDs are not aligned: only differences inside a block are used), block lifetime, resident rounds.  usage: tools/attn_trace.py [--stage 1..4]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import _lib, ops_swin as osw
p = argparse.ArgumentParser(); p.add_argument('--stage', type=int, default=0); a = p.parse_args()
L = _lib.lib()
dt = torch.bfloat16
NAMES = ['prologue (bias tile, zero sums, barrier)', 'operand loads -> registers -> LDS tiles', 'jb0: S, dP MFMAs + softmax + dS', 'jb0: four-wave fixed-order sum of dS (4 barriers)',
         'jb0: dQ MFMAs + store', 'jb1: S, dP MFMAs + softmax + dS', 'jb1: four-wave sum (4 barriers)', 'jb1: dQ MFMAs + store', 'N-layout: both key blocks (dK, dV)', 'final barrier + bias-gradient tile store']
for st, (hw, Cn, heads) in enumerate(((128, 96, 3), (64, 192, 6), (32, 384, 12), (16, 768, 24)), 1):
    if a.stage and a.stage != st: continue
    B, P = 8, osw.pad_to(Cn); P3 = osw.pad_to(3 * Cn)
    qkv = torch.randn(B, hw, hw, P3, device='cuda').to(dt)
    dout = torch.randn(B, hw, hw, P, device='cuda').to(dt)
    qb = torch.randn(3 * Cn, device='cuda'); rel = torch.randn(heads, 49, 49, device='cuda')
    for _ in range(3): osw.window_attention_bwd(qkv, qb, rel, dout, Cn, heads, 3)
    torch.cuda.synchronize()
    nblk = 16384
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device='cuda')
    L.sl_debug_attn_trace(ctypes.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); osw.window_attention_bwd(qkv, qb, rel, dout, Cn, heads, 3); e1.record()
    torch.cuda.synchronize()
    L.sl_debug_attn_trace(ctypes.c_void_p(0))
    t = buf.view(nblk, 16).cpu()
    t = t[t[:, 10] != 0]
    d = (t[:, 1:11] - t[:, 0:10]).double()
    life = (t[:, 10] - t[:, 0]).double()
    slots = 512          # two blocks per CU
    print('stage %d: %d tokens, %d heads: %d blocks = %.2f resident rounds of %d; traced launch %.1f us (the stamps cost 8 registers and 6 spills: the production launch is shorter);'
          ' block lifetime median %.0f shader ticks (min %.0f, max %.0f)' % (st, B * hw * hw, heads, t.shape[0], t.shape[0] / slots, slots, e0.elapsed_time(e1) * 1e3,
                                                                          float(life.median()), float(life.min()), float(life.max())))
    for i, nm in enumerate(NAMES):
        print('   %-62s median %7.0f ticks  (%4.1f %% of the block)' % (nm, float(d[:, i].median()), 100 * float(d[:, i].median()) / float(life.median())))
