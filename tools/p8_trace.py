#!/usr/bin/env python3
"""Where a conv_gemm_p8_kernel block spends its time: s_memtime stamps per block (entry, first K-tile landed, main loop done, end) + HW_ID,
through the debug hook sl_debug_p8_trace.  Prints mean phase lengths and, per CU, the gap between one block's end and the next block's entry."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops, _lib

p = argparse.ArgumentParser()
p.add_argument('--B', type=int, default=16); p.add_argument('--hw', type=int, default=64)
p.add_argument('--cin', type=int, default=512); p.add_argument('--cout', type=int, default=2048)
p.add_argument('--k', type=int, default=1); p.add_argument('--dil', type=int, default=1)
p.add_argument('--what', default='fwd')
a = p.parse_args()
dt = torch.bfloat16
spec = ops.ConvSpec(a.cin, a.cout, a.k, 1, a.dil * (a.k // 2), a.dil)
x = torch.randn(a.B, a.hw, a.hw, a.cin, device='cuda').to(dt)
w = torch.randn(a.cout, a.cin, a.k, a.k, device='cuda') * 0.02
wf, wb = ops.weight_prep(w, dt)
dy = torch.randn(a.B, a.hw, a.hw, a.cout, device='cuda').to(dt)
add = torch.randn(a.B, a.hw, a.hw, a.cin, device='cuda').to(dt)
bits = torch.randint(0, 256, (add.numel() // 8,), dtype=torch.uint8, device='cuda')
fn = {'fwd': lambda: ops.conv2d_fwd(x, wf, spec, want_stats=True),
      'dgrad': lambda: ops.conv2d_bwd_data(dy, wb, spec, (a.hw, a.hw)),
      'dgrad+add+bits': lambda: ops.conv2d_bwd_data(dy, wb, spec, (a.hw, a.hw), addend=add, addend_mask=bits)}[a.what]
for _ in range(3): fn()
M = a.B * a.hw * a.hw
N = a.cout if a.what == 'fwd' else a.cin
nblk = (M // 256) * (N // 256)
buf = torch.zeros(nblk, 8, dtype=torch.int64, device='cuda')
L = _lib.lib()
L.sl_debug_p8_trace(ctypes.c_void_p(buf.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
L.sl_debug_p8_trace(ctypes.c_void_p(0))
t = buf.cpu().numpy()
import numpy as np, collections
ms = e0.elapsed_time(e1)
xcc = t[:, 5] & 0xf
hw = t[:, 4]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7          # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]; s_memtime is per XCD (unsynchronised)
print('%s %d->%d k%d: %d blocks, launch (HIP events) %.1f us' % (a.what, a.cin, a.cout, a.k, nblk, ms * 1e3))
pro, main, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
tot = pro + main + epi
print('per block, shader ticks: prologue %.0f  main loop %.0f  epilogue %.0f  total %.0f   (%.1f %% / %.1f %% / %.1f %%)' %
      (pro.mean(), main.mean(), epi.mean(), tot.mean(), 100 * pro.mean() / tot.mean(), 100 * main.mean() / tot.mean(), 100 * epi.mean() / tot.mean()))
by = collections.defaultdict(list)
for i in range(nblk):
    by[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append((t[i, 0], t[i, 3]))
gaps, spans, busy = [], [], []
for k, v in by.items():
    v.sort()
    spans.append(v[-1][1] - v[0][0]); busy.append(sum(e - s_ for s_, e in v))
    for (s0, e0_), (s1, e1_) in zip(v, v[1:]):
        gaps.append(s1 - e0_)
g = np.array(gaps)
print('CU slots seen: %d; blocks per slot %.1f; per slot: first entry -> last end %.0f ticks (=> %.0f ticks/us), inside blocks %.1f %%' %
      (len(by), nblk / max(len(by), 1), np.mean(spans), np.mean(spans) / (ms * 1e3), 100 * np.sum(busy) / np.sum(spans)))
if len(g):
    print('gap between a block end and the next block entry on the same CU, ticks: median %.0f  mean %.0f  p10 %.0f  p90 %.0f' % (np.median(g), g.mean(), np.percentile(g, 10), np.percentile(g, 90)))
