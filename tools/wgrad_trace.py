#!/usr/bin/env python3
"""Where a weight-gradient block spends its time: s_memtime stamps per block (entry, ring primed, main loop done, slab stored) + HW_ID through the debug hooks
sl_debug_wgrad_trace (conv_wgrad_glds_kernel, one block per tap) and sl_debug_wgrad3_trace (conv_wgrad3_kernel, nine taps per block).  Prints mean phase lengths,
ticks per 32-row stage / per 16-pixel step, and per CU the share of the launch spent inside blocks.  --both: the two kernels on the same shape, one after the other."""
import argparse, collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from segland_amd import ops, _lib

p = argparse.ArgumentParser()
p.add_argument('--B', type=int, default=16); p.add_argument('--hw', type=int, default=64)
p.add_argument('--cin', type=int, default=256); p.add_argument('--cout', type=int, default=256)
p.add_argument('--k', type=int, default=3); p.add_argument('--dil', type=int, default=2)
p.add_argument('--both', action='store_true')
a = p.parse_args()
dt = torch.bfloat16
spec = ops.ConvSpec(a.cin, a.cout, a.k, 1, a.dil * (a.k // 2), a.dil)
x = torch.randn(a.B, a.hw, a.hw, a.cin, device='cuda').to(dt)
dy = torch.randn(a.B, a.hw, a.hw, a.cout, device='cuda').to(dt)
L = _lib.lib()
d = ops.conv_desc(dt, a.B, a.hw, a.hw, spec, None)
gf = 2.0 * a.B * a.hw * a.hw * a.cin * a.cout * a.k * a.k / 1e9


def run(nine):
    L.sl_debug_wgrad3(1 if nine else 0)
    cfg = L.sl_conv2d_wgrad_config(ctypes.byref(d))
    if nine and cfg != 3:
        print('shape not served by the nine-tap kernel (config %d)' % cfg); return
    if not nine and cfg < 10000000:
        print('shape not on conv_wgrad_glds_kernel (config %d)' % cfg); return
    fn = lambda: ops.conv2d_bwd_weight(x, dy, spec)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms_plain = e0.elapsed_time(e1) / 10
    nmax = 8192
    buf = torch.zeros(nmax, 8, dtype=torch.int64, device='cuda')
    hook = L.sl_debug_wgrad3_trace if nine else L.sl_debug_wgrad_trace
    hook(ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    hook(ctypes.c_void_p(0))
    ms = e0.elapsed_time(e1)
    t = buf.cpu().numpy()
    t = t[t[:, 3] != 0]
    nblk = len(t)
    name = 'conv_wgrad3_kernel' if nine else 'conv_wgrad_glds_kernel (config %d)' % cfg
    print('== %s  %d->%d k%d d%d B%d %dx%d: %d blocks; launch + reduce (HIP events) %.1f us untraced = %.0f TFLOP/s, %.1f us traced' %
          (name, a.cin, a.cout, a.k, a.dil, a.B, a.hw, a.hw, nblk, ms_plain * 1e3, gf / ms_plain, ms * 1e3))
    pro, main, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    tot = pro + main + epi
    nst = t[:, 6].astype(np.float64)
    print('per block, shader ticks: prologue %.0f  main loop %.0f  slab store %.0f  total %.0f   (%.1f %% / %.1f %% / %.1f %%)' %
          (pro.mean(), main.mean(), epi.mean(), tot.mean(), 100 * pro.mean() / tot.mean(), 100 * main.mean() / tot.mean(), 100 * epi.mean() / tot.mean()))
    if nine:
        print('main loop: %.1f stages of 4 steps per block, %.0f ticks per step (9 MFMAs per wave, 2 waves per SIMD: 576 MFMA-issue cycles)' % (nst.mean(), (main / (4 * nst)).mean()))
    else:
        print('main loop: %.1f stages of 32 rows per block, %.0f ticks per stage (256 x 256 tile: 1 024 MFMA-issue cycles per SIMD)' % (nst.mean(), (main / nst).mean()))
    xcc = t[:, 5] & 0xf; hw = t[:, 4]
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    by = collections.defaultdict(list)
    for i in range(nblk):
        by[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append((t[i, 0], t[i, 3]))
    spans, busy = [], []
    for k, v in by.items():
        v.sort(); spans.append(v[-1][1] - v[0][0]); busy.append(sum(e - s_ for s_, e in v))
    print('CU slots seen: %d; blocks per slot %.2f; per slot first entry -> last end %.0f ticks; launch incl. reduce %.0f ticks at 2.4 GHz' %
          (len(by), nblk / max(len(by), 1), np.mean(spans), ms * 1e3 * 2400))


run(False)
if a.both:
    run(True)
L.sl_debug_wgrad3(1)
