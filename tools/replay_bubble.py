#!/usr/bin/env python3
"""Does a HIP-graph replay carry a fixed bubble on this stack, and which ingredient of the training-step graph causes it?  (Kernel traces of the replayed ResNet-50 /
Swin-T steps show ~0.5 ms idle between the eager input copies in front of the replay and the graph's first node: 2 % / 5 % of the step.)  A chain of NK kernels that
each run ~20 us, replayed back to back; variants: plain; + an eager device-to-device copy in front of every replay (what GraphedStep does with the batch); + a captured
pinned-host-to-device copy node (what the optimizer's parameter table upload is); + both.  Wall time per replay minus the plain chain = the ingredient's cost."""
import sys, time
import torch
NK = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = 'cuda'
x = torch.randn(4 << 20, device=dev)            # 16 MB: a mul_ pass is ~10-20 us
src = torch.randn(16 << 20, device=dev); dst = torch.empty_like(src)
pinned = torch.zeros(4096, dtype=torch.uint8).pin_memory(); table = torch.zeros(4096, dtype=torch.uint8, device=dev)
def body(h2d):
    for _ in range(NK):
        x.mul_(1.0000001)
    if h2d:
        table.copy_(pinned, non_blocking=True)
        x.add_(table[0].float() * 0)
def capture(h2d):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(h2d)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(h2d)
    return g
def run(g, pre_copy, n=100):
    for _ in range(5):
        if pre_copy: dst.copy_(src, non_blocking=True)
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if pre_copy: dst.copy_(src, non_blocking=True)
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
g0, g1 = capture(False), capture(True)
base = run(g0, False)
print('%d-kernel chain, replayed back to back:                       %8.3f ms per replay' % (NK, base))
print('  + eager 64 MB device-to-device copy in front of the replay: %8.3f ms (copy alone ~0.03 ms)' % run(g0, True))
print('  + captured pinned-host -> device copy node (4 KB):          %8.3f ms' % run(g1, False))
print('  + both:                                                     %8.3f ms' % run(g1, True))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); 
for _ in range(20): body(False)
e1.record(); torch.cuda.synchronize()
print('the same chain issued kernel by kernel:                        %8.3f ms per pass' % (e0.elapsed_time(e1) / 20))
