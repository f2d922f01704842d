#!/usr/bin/env python3
"""Soak run of the drivers in ONE process (round 4): base training for several epochs with DataLoader workers decoding TIFF tiles + validation, a resumed run (-c), the
fine-tune driver for several epochs on both families, evaluation -- with the device / host memory after every stage, so that a leak or a stale-pointer fault that only
shows after many captures / loader restarts has a chance to show.  usage: python tools/soak.py [--epochs 6]"""
import argparse
import glob
import os
import resource
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def mem(tag, t0):
    torch.cuda.synchronize()
    print('SOAK %-28s %6.1f s   device allocated %7.1f MB  reserved %7.1f MB   host max RSS %6.0f MB' % (
        tag, time.time() - t0, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3), flush=True)


def main():
    p = argparse.ArgumentParser(); p.add_argument('--epochs', type=int, default=6); a = p.parse_args()
    from segland_amd import eval_base, ft_pop, graph_step, train_base
    tmp = tempfile.mkdtemp(prefix='soak_')
    t0 = time.time()
    for model, backbone in (('pspnet_pop', 'resnet50'), ('swin_pop', 'swin-t')):
        snap = os.path.join(tmp, model)
        common = ['--model', model, '--backbone', backbone, '--dataset', 'synthetic_tiff', '--input-size', '128,128', '--base-size', '160,160', '--print-frequency', '100',
                  '--num-workers', '3', '--restore-from', '/nonexistent', '--allow-random-init', '--fp16']
        train_base.main(common + ['--batch-size', '4', '--start-epoch', '36', '--num-epoch', str(36 + a.epochs), '--learning-rate', '1e-4', '--snapshot-dir', snap])      # validation from epoch 36 on (train_base.py:293): at epoch 40 and at the last one
        mem('%s train %d epochs' % (model, a.epochs), t0)
        state = sorted(glob.glob(os.path.join(snap, 'state_*.pth')))
        if state:
            train_base.main(common + ['--batch-size', '4', '--num-epoch', str(36 + a.epochs + 2), '--learning-rate', '1e-4', '--snapshot-dir', snap, '-c', state[-1]])
            mem('%s resumed +2 epochs' % model, t0)
        ck = sorted(glob.glob(os.path.join(snap, 'epoch_*.pth')))[-1]
        snap_ft = os.path.join(tmp, model + '_ft')
        ft_pop.main(['--model', model, '--backbone', backbone, '--dataset', 'synthetic_tiff', '--batch-size', '1', '--input-size', '128,128', '--base-size', '128,128',
                     '--num-epoch', str(a.epochs), '--learning-rate', '1e-3', '--print-frequency', '100', '--snapshot-dir', snap_ft, '--shot', '3', '--num-workers', '3',
                     '--restore-from', ck, '--random-seed', '123', '--freeze-backbone', '--update-base'])
        mem('%s ft_pop %d epochs' % (model, a.epochs), t0)
        res = eval_base.main(['--model', model, '--backbone', backbone, '--dataset', 'synthetic_tiff', '--base-size', '160,160', '--test-batch-size', '2', '--fp16',
                              '--restore-from', ck, '--num-workers', '2', '--save-path', os.path.join(tmp, model + '_eval')])
        mem('%s eval_base' % model, t0)
        assert all(torch.isfinite(torch.tensor(v[2])) for v in res.values()), res
    print('SOAK graph_step.STATS', graph_step.STATS)
    assert graph_step.STATS['failures'] == 0
    print('SOAK ok')


if __name__ == '__main__':
    main()
