#!/usr/bin/env python3
"""Where a conv_gemm_sk512_kernel block (wave 0) spends its time: s_memtime sums per phase over the steps, through the debug hook sl_debug_p8_trace."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segland_amd import ops, _lib
p = argparse.ArgumentParser()
p.add_argument('--B', type=int, default=16); p.add_argument('--hw', type=int, default=64); p.add_argument('--n', type=int, default=2048)
p.add_argument('--what', default='fwd')
a = p.parse_args()
dt = torch.bfloat16
K = 512
x = torch.randn(a.B, a.hw, a.hw, K, device='cuda').to(dt)
if a.what == 'fwd':
    spec = ops.ConvSpec(K, a.n, 1, 1, 0, 1)
    wf, _ = ops.weight_prep(torch.randn(a.n, K, 1, 1, device='cuda') * 0.05, dt)
    fn = lambda: ops.conv2d_fwd(x, wf, spec, want_stats=True)
else:
    spec = ops.ConvSpec(a.n, K, 1, 1, 0, 1)
    _, wb = ops.weight_prep(torch.randn(K, a.n, 1, 1, device='cuda') * 0.05, dt)
    add = torch.randn(a.B, a.hw, a.hw, a.n, device='cuda').to(dt)
    bits = torch.randint(0, 256, (add.numel() // 8,), dtype=torch.uint8, device='cuda')
    kw = {'dgrad': {}, 'dgrad+add': {'addend': add}, 'dgrad+add+bits': {'addend': add, 'addend_mask': bits}}[a.what]
    fn = lambda: ops.conv2d_bwd_data(x, wb, spec, (a.hw, a.hw), **kw)
for _ in range(3): fn()
M = a.B * a.hw * a.hw
buf = torch.zeros(M // 256, 8, 8, dtype=torch.int64, device='cuda')
L = _lib.lib()
L.sl_debug_p8_trace(ctypes.c_void_p(buf.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
L.sl_debug_p8_trace(ctypes.c_void_p(0))
tw = buf.cpu().numpy().astype('float64')          # [block][wave][8]
t = tw[:, 0, :]
names = ['prologue (A rows + first halves)', 'multiply (2 x 32 MFMAs) + staging', 'vmcnt(0) at the interval end', 'barrier', '-', 'store phase', '-', 'TOTAL']
ns = a.n // 64
print('%s 512 -> %d: %d blocks, %d steps each; launch (HIP events, traced) %.1f us' % (a.what, a.n, len(t), ns, e0.elapsed_time(e1) * 1e3))
for k, nm in enumerate(names):
    print('  %-36s %9.0f ticks  %5.1f %%   %s' % (nm, t[:, k].mean(), 100 * t[:, k].mean() / t[:, 7].mean(), '' if k in (0, 7) else '(%.0f per step)' % (t[:, k].mean() / ns)))
print('per wave (mean over blocks), ticks per step:  multiply / vmcnt wait / barrier / store')
for w in range(8):
    print('  wave %d: %6.0f %6.0f %6.0f %6.0f' % (w, tw[:, w, 1].mean() / ns, tw[:, w, 2].mean() / ns, tw[:, w, 3].mean() / ns, tw[:, w, 5].mean() / ns))
