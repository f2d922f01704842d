import os, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0); dist.init_process_group('nccl')
for mb in (0.001, 1, 16, 64):
    t = torch.zeros(int(mb * 2**20 / 4) or 1, device='cuda')
    for _ in range(5): dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dist.all_reduce(t)
    e1.record(); torch.cuda.synchronize()
    # async on side stream like DDP: launch, then make current stream wait
    t0 = time.perf_counter()
    for _ in range(20):
        w = dist.all_reduce(t, async_op=True); w.wait()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('%.3f MB: %.1f us per all_reduce (stream time), %.1f us per async+wait round trip (host)' % (mb, 1e3 * e0.elapsed_time(e1) / 20, 1e6 * (t1 - t0) / 20))
dist.destroy_process_group()
