#!/usr/bin/env python3
"""What starting DataLoader workers costs in a process that has initialised the GPU: `fork` (torch's default on Linux) duplicates the parent's address space -- with the
HIP runtime's mappings in it -- per worker and per epoch; `forkserver` forks from a small server process that has torch imported.  usage: tools/loader_start.py [--workers 2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    p = argparse.ArgumentParser(); p.add_argument('--workers', type=int, default=2); p.add_argument('--gb', type=float, default=8.0)
    a = p.parse_args()
    from segland_amd.dataset import synthetic_raw
    ds = synthetic_raw.GFSSegTrain(crop_size=(128, 128), length=16)
    from segland_amd.dataset.oem import raw_collate

    def start(ctx):
        t0 = time.perf_counter()
        dl = torch.utils.data.DataLoader(ds, batch_size=4, num_workers=a.workers, collate_fn=raw_collate, multiprocessing_context=ctx)
        it = iter(dl); next(it)
        t1 = time.perf_counter()
        for _ in it:
            pass
        del it, dl
        return t1 - t0
    print('before any GPU call: fork %.2f s' % start('fork'))
    x = torch.zeros(int(a.gb * (1 << 30)), dtype=torch.uint8, device='cuda'); torch.cuda.synchronize()
    from segland_amd import ops  # noqa: F401
    print('after GPU init + %.0f GB allocated: fork %.2f s, again %.2f s' % (a.gb, start('fork'), start('fork')))
    import multiprocessing as mp
    mp.set_forkserver_preload(['torch', 'segland_amd.dataset.synthetic_raw'])
    print('forkserver: first %.2f s (starts the server), again %.2f s, again %.2f s' % (start('forkserver'), start('forkserver'), start('forkserver')))
    print('spawn: %.2f s' % start('spawn'))


if __name__ == '__main__':
    main()
