"""Round-3 gates: data parallelism a HIP graph can hold (bucket_step.py: two graphs around the bucket all-reduces), the kernels and test
holes the round-2 review named (the 3x3 patch kernel on the bench's big shapes against fp32 torch in isolation, the 512x512 batch-16 bf16
train-mode gate, Swin-T at the bench shape)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import formula as fm

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


# --------------------------------------------------------------------------------------------- N > 1 without DistributedDataParallel
@pytest.mark.timeout(600)
def test_bucket_step_under_rccl_world1(hip):
    """A fresh child process, world_size-1 RCCL group: Engine.data_parallel(graphable=True) returns a BucketedReplica, GraphedBucketStep replays
    graph A (forward + backward into the build's gradient buckets) / RCCL all-reduce per bucket / graph B (clip + AdamW) -- five iterations
    must leave parameters, buffers, AdamW-driven losses and eval logits identical to the unwrapped kernel-by-kernel run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ddp_child.py'), 'bucket', str(_free_port())], env=env, capture_output=True,
                       text=True, timeout=540)
    line = [l for l in r.stdout.splitlines() if l.startswith('DDP_CHILD ')]
    assert r.returncode == 0 and line, 'child failed (rc %d):\n%s\n%s' % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    out = json.loads(line[-1][len('DDP_CHILD '):])
    print(out)
    assert out['bucket_failures'] == 0 and out['bucket_replays'] >= 3 and out['buckets'] >= 2, out
    assert out['grads_alias_cached_views'] >= 175, out               # every gradient lives in a bucket after the step
    assert out['bn1_tracked'] == out['ref_bn1_tracked'] == 5
    assert out['worst_param_rel'] <= 1e-6 and out['logits_rel'] <= 1e-6, out
    for (a, ga), (b, gb) in zip(out['losses'], out['ref_losses']):
        assert abs(a - b) <= 1e-6 * abs(b) and abs(ga - gb) <= 1e-5 * gb


@pytest.mark.timeout(1500)
def test_bucket_step_two_ranks_equals_ddp(hip, tmp_path):
    """Two ranks on the one GPU of the box (gloo), a different half-batch per rank and iteration, per-GPU BatchNorm statistics: the two-graph
    bucket step must train exactly like DistributedDataParallel with in-place bucket gradients (round 2's path, itself pinned against the
    one-process full batch by test_two_ranks_equal_one_full_batch): same losses, same parameters on both ranks and in both modes."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    child = os.path.join(ROOT, 'tests', 'bucket2_child.py')
    res = {}
    for mode in ('ddp', 'bucket'):
        port, out = str(_free_port()), str(tmp_path / (mode + '.pt'))
        procs = [subprocess.Popen([sys.executable, child, str(r), port, out, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
        logs = [p.communicate(timeout=700)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), mode + ' rank failed:\n' + '\n----\n'.join(l[-3000:] for l in logs)
        res[mode] = torch.load(out)
    d, b = res['ddp'], res['bucket']
    print('losses ddp   :', d['losses'], '\nlosses bucket:', b['losses'], '\nbuckets', b['buckets'], 'replays', b['replays'])
    assert d['ranks_equal'] and b['ranks_equal']
    assert b['replays'] >= 2 * 3 and b['buckets'] >= 2
    for (x, gx), (y, gy) in zip(d['losses'], b['losses']):
        assert abs(x - y) <= 1e-6 * abs(y) and abs(gx - gy) <= 1e-5 * gy
    worst, key = 0.0, ''
    for k, v in d['sd'].items():
        e = float((b['sd'][k] - v).abs().max() / max(float(v.abs().max()), 1e-12))
        if e > worst:
            worst, key = e, k
    print('worst parameter / buffer difference %.3e (%s)' % (worst, key))
    assert worst <= 1e-5, (worst, key)
    assert float((d['logits'] - b['logits']).abs().max() / d['logits'].abs().max()) <= 1e-5
