#!/usr/bin/env python3
"""Headline benchmark: 512x512 tiles/sec, forward+backward (+clip +AdamW), PSPNet-POP on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: one rank per GPU -- under torch.distributed.run, or started by bench.py itself
                                                          as fresh child processes when no launcher set WORLD_SIZE)

A "step" is the loop body of the reference's train_base.py:250-264 over one synthetic batch that is already resident
in HBM: zero_grad, forward (backbone + PPM + POP head + fused upsample/CE + orth loss), backward, clip_grad_norm_(5.0),
AdamW step(s).  Workload at every N: BASELINE.json configs[1] per GPU -- PSPNet-POP ResNet-50, bf16, batch 16, 512x512,
8 logit channels -- so scaling is weak and `value` is the whole-job tiles/s.

The JSON line also carries
  roofline     : the dominant kernel (picked by an instrumented warm-up pass), timed live with HIP events on the
                 launch stream during the timed steps; achieved = algorithmic FLOPs of that launch / mean duration,
                 peak = 2500 TFLOP/s (bf16 dense MFMA, MI355X_MICROARCH.md);
  cpu_baseline : the CPU oracle (a port: oracle/pop_oracle.py) running the same loop body on config C1 (batch 2,
                 fp32) on this host's cores -- rank 0, N = 1 only.  Reported, not a target.
"""
import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_TILE = {'resnet50': 1170.7, 'resnet101': 1636.1, 'swin-t': 197.4}      # fwd+bwd, reference algorithm
PEAK_BF16_TFLOPS = 2500.0
PEAK_F32_TFLOPS = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--model', default='pspnet_pop', choices=['pspnet_pop', 'swin_pop'], help='swin_pop: BASELINE config 5 (Swin-T + UperNet_Decoder_Plus + POP head, 8 tiles per GPU)')
    p.add_argument('--batch', type=int, default=None, help='tiles per GPU (default 16; swin_pop: 8)')
    p.add_argument('--backbone', default=None)
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    p.add_argument('--size', type=int, default=512)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-budget', type=float, default=40.0, help='seconds of CPU work allowed for the cpu_baseline sample')
    p.add_argument('--torch-optimizer', action='store_true', help='torch.optim.AdamW(fused=True) instead of segland_amd.optim.AdamW')
    p.add_argument('--single-step', action='store_true', help='one AdamW step per iteration (the reference does two, train_base.py:262-264)')
    p.add_argument('--no-step-graph', action='store_true', help='issue the timed steps kernel by kernel (default on one GPU: they replay the step as ONE HIP graph, segland_amd/graph_step.py, '
                   'like train_base does).  Either way a second region of instrumented kernel-by-kernel steps carries the roofline events')
    p.add_argument('--profile-table', default='', help='write a per-kernel-shape timing table (instrumented extra pass) to this file')
    p.add_argument('--side-config', default='', choices=['', 'c3', 'c4', 'c5'], help='internal: measure ONE of the other BASELINE configurations and print its JSON object (the default N = 1 run '
                   'starts one fresh child process per configuration, so that a fault in a side measurement cannot take the headline line with it)')
    p.add_argument('--no-other-configs', action='store_true', help='skip the short measurements of BASELINE configs 3 (ResNet-101 shard), 4 (fine-tune pair) and 5 (Swin-T shard) '
                   'that ride on the default N = 1 run (`other_configs` in the JSON line); profiling runs pass this so that the trace holds the headline workload only')
    return p.parse_args()


def synthetic_batch(B, size, device, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    img = torch.randn(B, 3, size, size, generator=g)
    mask = torch.randint(0, 8, (B, size, size), generator=g, dtype=torch.int64)
    mask[0, :50] = 255
    return img.to(device), mask.to(device)


def make_optimizer(model, lr=1e-3, wd=1e-4, torch_optimizer=False):
    from segland_amd.utils.pyt_utils import get_parameters
    if torch_optimizer:
        return torch.optim.AdamW(get_parameters(model, lr=lr), lr=lr, weight_decay=wd, fused=True)
    from segland_amd.optim import AdamW          # same update rule and state_dict, one launch over all parameters (csrc/optim.hip)
    return AdamW(get_parameters(model, lr=lr), lr=lr, weight_decay=wd)


def train_step(model, opt, img, mask, params, double_step, grad_div=1):
    opt.zero_grad(set_to_none=True)
    loss = model(img, mask)
    loss['total_loss'].backward()
    from segland_amd import ops
    # the optimizer part of the step (reported separately as optimizer_ms_per_step): gradient norm (4 B per parameter) + AdamW (read p, g, m, v; write p, m, v: 28 B)
    with ops.PROFILER.region('optimizer (clip_grad_norm_ + AdamW)', 32 * sum(p.numel() for p in params) if ops.PROFILER.on else 0):
        if hasattr(opt, 'repeat_next'):
            from segland_amd.optim import clip_coefficient
            _, coef = clip_coefficient(params, 5.0, grad_div)   # clip_grad_norm_(5.0): the coefficient is applied inside the optimizer kernel
            opt.step(repeat=2 if double_step else 1, grad_scale=coef)      # the reference's two steps on the same gradients in one pass
        else:
            torch.nn.utils.clip_grad_norm_(params, 5.0)
            opt.step()
            if double_step:
                opt.step()
    return loss


def usable_cpus():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (containers report all host CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(budget_s, backbone, model='pspnet_pop'):
    """CPU port (oracle) of the same loop body, on a BOUNDED sample: batch 2 at 256x256 first (1/4 of the pixels of
    config C1); if that step takes < budget/8 the full C1 sample (batch 2, 512x512) is timed too and reported instead.
    tiles/s is always in units of 512x512 tiles (a 256x256 tile counts 1/4)."""
    from oracle import pop_oracle as po
    torch.manual_seed(0)
    threads = usable_cpus()
    torch.set_num_threads(threads)
    if model == 'swin_pop':
        from oracle import swin_oracle as so
        m = so.SwinPopOracle(7, criterion=po.OrthLossOracle(255), backbone=backbone).train()
    else:
        m = po.PopOracle(n_base=7, criterion=po.OrthLossOracle(255), backbone=backbone).train()
    groups, _ = po.param_groups(m, lr=1e-5)
    opt = torch.optim.AdamW(groups, lr=1e-5, weight_decay=1e-4)

    def timed(size, n):
        img, mask = synthetic_batch(2, size, 'cpu')
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            po.train_step(m, opt, img, mask)
            ts.append(time.perf_counter() - t0)
        return ts

    # 1 warm-up + >= 3 timed steps, median
    def median(ts):
        ts = sorted(ts)
        return ts[len(ts) // 2]
    small = timed(256, 4)
    size, t, n_timed = 256, median(small[1:]), 3        # first call is the warm-up
    if t * 4 * 4 < budget_s:                            # the full C1 sample (batch 2 at 512x512): 1 warm-up + 3 timed steps fit the budget
        full = timed(512, 4)
        size, t = 512, median(full[1:])
    cpu = ''
    try:
        cpu = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:
        pass
    return {'value': 2.0 * (size * size) / (512.0 * 512.0) / t, 'unit': 'tiles/s', 'cores': threads, 'kind': 'port',
            'sample': ('oracle/swin_oracle.py' if model == 'swin_pop' else 'oracle/pop_oracle.py') + ' train_step (train_base.py:250-264 body), %s fp32, batch 2 at %dx%d, median of 3 timed steps after 1 warm-up: %.2f s/step, '
                      '%d threads (cgroup/affinity limit) on %s' % (backbone, size, size, t, threads, cpu)}


def visible_gpus():
    """GPUs a child rank would see, counted WITHOUT a HIP call in this process: the KFD topology nodes that have SIMDs, cut down by
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when the topology is not readable (the pre-check is then skipped and
    a wrong rank count fails in the children).  torch.cuda.device_count() is NOT used: on ROCm without amdsmi it calls hipGetDeviceCount, which brings the
    HIP / HSA runtime up in the parent (round-3 advisor)."""
    import glob
    if not os.path.isdir('/sys/class/kfd/kfd/topology/nodes'):
        return 0                                            # no KFD driver on this host: no AMD GPU for any child either
    n = 0
    try:
        for f in glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties'):
            for line in open(f):
                if line.startswith('simd_count') and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return None
    if n == 0:
        return None
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def lib_sha16():
    """First 16 hex digits of the sha256 of the loaded libsegland_hip.so (what the committed counter files are keyed on)."""
    import hashlib
    from segland_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def _dominant(table, peak):
    """(kernel family with the most time in an instrumented step, its fraction of `peak` TFLOP/s, its ms per step)."""
    if not table:
        return None, None, None
    e = max(table.values(), key=lambda t: t['ms_total'])
    return e['family'], round(e['gflop'] / max(e['ms_total'], 1e-9) / peak, 4), round(e['ms_total'], 3)


def short_config(kind, dev, steps=None):
    """One of BASELINE.json's other single-GPU workloads, measured briefly in this process after the headline region so that the driver's N = 1 line shows them:
    'c3' = config 3's per-GPU shard (PSPNet-POP ResNet-101, bf16, 16 tiles), 'c5' = config 5's (Swin-T POP, bf16, 8 tiles), both the train_base.py loop body
    replayed as one HIP graph; 'c4' = config 4 (ft_pop.py:233-269: one novel + one base tile per step, frozen backbone + decoder, SGD on the novel head).
    3 eager warm-up steps (the last one instrumented: dominant conv kernel family and its fraction of the 2 500 TFLOP/s peak), graph capture, `steps` timed replays
    (default: 20 / 200 / 40 for c3 / c4 / c5)."""
    from segland_amd import graph_step, networks, ops
    from segland_amd.loss.criterion import OrthLoss
    torch.manual_seed(0)
    dt = torch.bfloat16
    if steps is None:
        steps = {'c3': 20, 'c4': 200, 'c5': 40}[kind]          # >= 0.4 s of timed replays each (10 steps of the 2 ms pair step read 2-3 % low against tools/bench_ft.py)
    if kind in ('c3', 'c5'):
        name, backbone, batch = ('pspnet_pop', 'resnet101', 16) if kind == 'c3' else ('swin_pop', 'swin-t', 8)
        kw = dict(dilated=True, os=8) if name == 'pspnet_pop' else {}
        with contextlib.redirect_stdout(sys.stderr):
            model = getattr(networks, name).GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone=backbone, pretrained_model=None, compute_dtype=dt, **kw).to(dev).train()
        opt = make_optimizer(model)
        params = [p for p in model.parameters() if p.requires_grad]
        batches = [synthetic_batch(batch, 512, dev, seed=100 + k) for k in range(2)]
        for i in range(3):
            if i == 2:
                ops.PROFILER.start()
            train_step(model, opt, *batches[i % 2], params, True)
        torch.cuda.synchronize()
        table = ops.PROFILER.stop(); ops.PROFILER.stop_bytes()
        fn = lambda m_, o_, s_, im_, mk_, double_step=True: (train_step(m_, o_, im_, mk_, params, double_step), None)      # noqa: E731
        graphed = graph_step.GraphedTrainStep(fn, model, opt, None, double_step=True, warmup=0)
        for k in range(3):
            graphed(*batches[k % 2])
        step = (lambda k: graphed(*batches[k % 2])) if graphed.graph is not None else (lambda k: train_step(model, opt, *batches[k % 2], params, True))
        units, unit, gflop = batch, 'tiles/s', GFLOP_PER_TILE[backbone]
        issue = 'one HIP graph replay per step' if graphed.graph is not None else 'kernel by kernel'
        what = '%s %s bf16, batch %d, 512x512, train_base.py loop body (BASELINE config %s per GPU)' % ('PSPNet-POP' if kind == 'c3' else 'Swin-POP', backbone, batch, kind[1])
    else:
        from segland_amd.ft_pop import ft_graph_body, ft_iteration, ft_iteration_graphed
        from segland_amd.utils.pyt_utils import NativeScalerWithGradNormCount, get_parameters
        with contextlib.redirect_stdout(sys.stderr):
            model = networks.pspnet_pop.GFSS_Model(n_base=7, criterion=OrthLoss(255), is_ft=True, n_novel=4, backbone='resnet50', pretrained_model=None, compute_dtype=dt,
                                                   dilated=True, os=8).to(dev)
        model.init_cls_n()
        from segland_amd.optim import SGD                 # torch.optim.SGD semantics, one capturable launch (what ft_pop uses on the GPU)
        opt = SGD(get_parameters(model, lr=1e-3, freeze_backbone=True), lr=1e-3, momentum=0.9, weight_decay=5e-4)
        g = torch.Generator(device='cpu').manual_seed(7)
        img, img_b = torch.randn(1, 3, 512, 512, generator=g).to(dev), torch.randn(1, 3, 512, 512, generator=g).to(dev)
        mask = torch.randint(8, 12, (1, 512, 512), generator=g).to(dev); mask[:, :40] = 255
        mask_b = torch.randint(0, 8, (1, 512, 512), generator=g).to(dev)
        model.train_mode()
        sc = NativeScalerWithGradNormCount()
        from segland_amd.networks import pspnet_pop as _pp
        fg = _pp._FEATURE_GRAPH
        for i in range(3):
            if i == 2:
                ops.PROFILER.start()
                _pp._FEATURE_GRAPH = False                  # the instrumented step issues the frozen feature extractor kernel by kernel (a graph replay cannot carry event pairs)
            try:
                ft_iteration(model, opt, sc, (img, mask, img_b, mask_b.clone()), dev)
            finally:
                _pp._FEATURE_GRAPH = fg
        torch.cuda.synchronize()
        table = ops.PROFILER.stop(); ops.PROFILER.stop_bytes()
        graphed = graph_step.GraphedStep(ft_graph_body(model, optimizer=opt), model, opt) if graph_step.eligible(model, opt, dev, need_adamw=False) else None
        if graphed is not None:
            for k in range(5):
                ft_iteration_graphed(graphed, opt, (img, mask, img_b, mask_b), dev)
        replayed = graphed is not None and graphed.graph is not None
        step = (lambda k: ft_iteration_graphed(graphed, opt, (img, mask, img_b, mask_b), dev)) if replayed else (lambda k: ft_iteration(model, opt, sc, (img, mask, img_b, mask_b.clone()), dev))
        units, unit, gflop = 1, 'pairs/s', 901.7                                      # the pair as the reference executes it
        issue = 'forward + backward + clip + SGD step replayed as one HIP graph' if replayed else 'kernel by kernel'
        what = 'ft_pop.py novel-class update: 1 novel + 1 base 512x512 tile per step, PSPNet-POP ResNet-50 bf16, frozen backbone + decoder (BASELINE config 4)'
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    value = units * steps / dt_s
    fam, frac, fam_ms = _dominant(table, PEAK_BF16_TFLOPS)
    return {'workload': what, 'value': round(value, 2), 'unit': unit, 'ms_per_step': round(1e3 * dt_s / steps, 3), 'steps': steps, 'step_issue': issue,
            'whole_step_frac_of_peak': round(value * gflop / 1e3 / PEAK_BF16_TFLOPS, 4), 'reference_gflop_per_unit': gflop,
            'dominant_kernel': fam, 'dominant_kernel_frac_of_peak': frac, 'dominant_kernel_ms_per_step': fam_ms}


def self_launch(n, argv=None):
    """`python bench.py --gpus N` as a plain command (no launcher): start the N ranks as FRESH child processes of
    `python -m torch.distributed.run` and pass rank 0's JSON line through.  Returns the exit code.  The parent makes no GPU call at all -- a process that
    initialised the GPU must never exec or become a rank, and it would keep a runtime handle open beside the N ranks: the devices are counted from
    sysfs (visible_gpus)."""
    import socket
    import subprocess
    have = visible_gpus()
    if have is not None and have < n:
        print('bench.py --gpus %d: this node exposes %d GPU(s); one rank per GPU is required (RCCL refuses two ranks on one device). '
              'Run with --gpus %d or fewer.' % (n, have, max(have, 1)), file=sys.stderr, flush=True)
        return 3
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    print('bench: --gpus %d without a launcher: starting %d ranks as child processes: %s' % (n, n, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.batch is None:
        a.batch = 8 if a.model == 'swin_pop' else 16
    if a.backbone is None:
        a.backbone = 'swin-t' if a.model == 'swin_pop' else 'resnet50'
    if a.side_config:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
        torch.cuda.set_device(0)
        print(json.dumps(short_config(a.side_config, torch.device('cuda', 0))), flush=True)
        return
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != a.gpus and world > 1:
        a.gpus = world
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_ddp = world > 1 or os.environ.get('SEGLAND_FORCE_DDP') == '1'       # the env knob exercises DDP/RCCL on one GPU
    if use_ddp:
        dist.init_process_group('nccl', init_method='env://')       # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == max(world, 1), 'process group has %d ranks, launcher said %d' % (dist.get_world_size(), world)
    assert a.gpus == world or world == 1, '--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run --nproc-per-node %d' % (a.gpus, world, a.gpus)
    if rank == 0:
        print('bench: world size %d (%s), rank 0 on cuda:%d' % (world, 'RCCL/DDP' if use_ddp else 'single process', local), file=sys.stderr, flush=True)

    from segland_amd import ops
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd import networks

    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    torch.manual_seed(0)
    GFSS_Model = getattr(networks, a.model).GFSS_Model
    kw = dict(dilated=True, os=8) if a.model == 'pspnet_pop' else {}
    with contextlib.redirect_stdout(sys.stderr):            # the factory prints 'Backbone:<name>' like the reference's; stdout carries the JSON line only
        model = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone=a.backbone, pretrained_model=None, compute_dtype=dtype, **kw).to(dev).train()
    opt = make_optimizer(model, torch_optimizer=a.torch_optimizer)
    net = model
    grad_div = 1
    from segland_amd import bucket_step
    replica = None
    if use_ddp and not a.torch_optimizer and not a.no_step_graph and bucket_step.eligible(world, True):
        # N > 1 default: gradient buckets owned by the build -- graph A (forward + backward into the buckets), one RCCL all-reduce per bucket,
        # graph B (clip + AdamW): three host actions per step instead of ~750 launches under DistributedDataParallel's reducer (bucket_step.py)
        net = replica = bucket_step.BucketedReplica(model, cap_mb=64)
        grad_div = world
    elif use_ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], broadcast_buffers=False,
                                                        gradient_as_bucket_view=True, bucket_cap_mb=64)
        if not a.torch_optimizer and os.environ.get('SEGLAND_DDP_PLAIN') != '1':
            from segland_amd.engine import enable_inplace_bucket_gradients      # sum-only all-reduce, gradients written into the bucket views
            enable_inplace_bucket_gradients(net)
            grad_div = world
    params = [p for p in model.parameters() if p.requires_grad]
    batches = [synthetic_batch(a.batch, a.size, dev, seed=rank * 16 + k) for k in range(4)]      # four resident batches, cycled: no step sees the previous one's tiles
    img, mask = batches[0]
    double = not a.single_step

    eager_fn = ((lambda img, mask: replica.train_iteration(opt, img, mask, double)) if replica is not None      # noqa: E731
                else (lambda img, mask: train_step(net, opt, img, mask, params, double, grad_div)))
    # ---- warm-up; the last warm-up step is instrumented to find the dominant kernel shape
    for i in range(a.warmup):
        if i == a.warmup - 1:
            ops.PROFILER.start()
        eager_fn(img, mask)
    if a.warmup == 0:
        ops.PROFILER.start()
        eager_fn(img, mask)
    torch.cuda.synchronize()
    table = ops.PROFILER.stop()
    table_bytes = ops.PROFILER.stop_bytes()
    # the kernel with the most time in the step, whichever family it is (in this instrumented pass a weight-gradient span includes its small fixed-order
    # slab reduce)
    dominant = max(table.values(), key=lambda e: e['ms_total']) if table else None

    # ---- the product's default on one GPU: the whole step replayed as ONE HIP graph (train_base does the same).  A graph replay cannot carry
    # per-launch events, so the K timed steps that give `value` are replays and a SECOND region of K kernel-by-kernel steps right after it
    # carries the event pairs of the roofline kernel (same kernels, same shapes, same process).
    from segland_amd import graph_step
    graphed = None
    if replica is not None:
        graphed = bucket_step.GraphedBucketStep(replica, opt, double_step=double, warmup=0)
        for k in range(3):
            graphed(*batches[k % len(batches)])
        if graphed.graph is None:
            graphed = None
    elif not a.no_step_graph and not use_ddp and graph_step.eligible(net, opt, dev):
        fn = lambda m_, o_, s_, im_, mk_, double_step=True: (train_step(m_, o_, im_, mk_, params, double_step, grad_div), None)      # noqa: E731
        graphed = graph_step.GraphedTrainStep(fn, net, opt, None, double_step=double, warmup=0)
        for k in range(3):                                 # capture + two untimed replays
            graphed(*batches[k % len(batches)])
        if graphed.graph is None:
            graphed = None

    def timed_region(step_fn):
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(a.steps):
            img, mask = batches[k % len(batches)]
            step_fn(img, mask)
            marks[k + 1].record()                          # per-step GPU time without a host synchronisation (median below)
        torch.cuda.synchronize()
        if use_ddp:
            dist.barrier()
        return time.perf_counter() - t0, marks

    # Region 1 -- the K timed steps of `value`: uninstrumented (graph replays on one GPU; kernel by kernel under DDP).
    # Region 2 -- K more steps issued kernel by kernel with HIP-event pairs around every launch of the roofline kernel (each pair costs two
    # ~5.7 us queue markers: 0.3 ms per R50 step, more next to DDP's RCCL stream, so they stay out of the number the driver compares).
    dt_s, marks = timed_region((lambda img, mask: graphed(img, mask)) if graphed is not None else eager_fn)
    eager_ms = None
    if os.environ.get('SEGLAND_BENCH_NOEVENTS') != '1':
        if dominant is not None:
            ops.PROFILER.start(only=dominant['family'])
        dt_e, _ = timed_region(eager_fn)
        eager_ms = 1e3 * dt_e / a.steps
    live = ops.PROFILER.stop()
    if use_ddp:
        t = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())

    if a.profile_table and rank == 0:
        with open(a.profile_table, 'w') as f:
            f.write('# one instrumented step, %s %s batch %d: conv launches by kernel and shape (HIP events on the launch stream)\n' % (a.backbone, a.dtype, a.batch))
            for fam in sorted(table.values(), key=lambda e: -e['ms_total']):
                f.write('== %-58s calls %4d  %8.3f ms  %9.1f GFLOP  %7.1f TFLOP/s  %6.2f TB/s   floor %7.3f ms (%.2f of it)\n' % (
                    fam['family'], fam['calls'], fam['ms_total'], fam['gflop'], fam['gflop'] / max(fam['ms_total'], 1e-9), fam.get('gbytes', 0.0) / max(fam['ms_total'], 1e-9),
                    fam['floor_ms'], fam['floor_ms'] / max(fam['ms_total'], 1e-9)))
                for shape, (n, ms, gf, fl) in sorted(fam['shapes'].items(), key=lambda kv: -kv[1][1]):
                    f.write('   %-50s %4d %9.3f ms %9.1f GFLOP %8.1f TFLOP/s   floor %7.3f ms (%.2f)\n' % (shape, n, ms, gf, gf / max(ms, 1e-9), fl, fl / max(ms, 1e-9)))
            f.write('# total conv time %.2f ms, total conv GFLOP %.1f, total conv floor %.2f ms\n' % (sum(e['ms_total'] for e in table.values()), sum(e['gflop'] for e in table.values()),
                                                                                                        sum(e['floor_ms'] for e in table.values())))
            f.write('# floor of a launch = max(FLOP / dense MFMA peak (2500 TFLOP/s bf16, 157.3 fp32), algorithmic operand + result bytes / 6.3 TB/s achievable HBM); a weight-gradient span includes its slab reduce\n')
            f.write('# streaming families of the same step (algorithmic bytes; floor = bytes / 6.3 TB/s):\n')
            for k_, e in sorted(table_bytes.items(), key=lambda kv: -kv[1]['ms_total']):
                f.write('== %-58s calls %4d  %8.3f ms  %9.3f GB  %6.2f TB/s   floor %7.3f ms (%.2f of it)\n' % (k_, e['calls'], e['ms_total'], e['gbytes'], e['gbytes'] / max(e['ms_total'], 1e-9),
                                                                                                              e['floor_ms'], e['floor_ms'] / max(e['ms_total'], 1e-9)))

    if rank == 0:
        tiles = a.batch * world * a.steps
        value = tiles / dt_s
        peak = PEAK_BF16_TFLOPS if a.dtype == 'bf16' else PEAK_F32_TFLOPS
        out = {
            'metric': '512x512 tiles/sec fwd+bwd ' + ('PSPNet-POP' if a.model == 'pspnet_pop' else 'Swin-POP'), 'value': round(value, 3), 'unit': 'tiles/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt_s / a.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': ('PSPNet-POP' if a.model == 'pspnet_pop' else 'Swin-POP') + ' %s %s, batch %d/GPU, %dx%d, 8 logit channels, train_base.py loop body '
                                   '(fwd+loss+bwd+clip+AdamW x%d), %d x MI355X' % (a.backbone, a.dtype, a.batch, a.size, a.size, 2 if double else 1, world),
                       'global_batch': a.batch * world, 'parallelism': 'dp%d' % world},
            'whole_step_tflops': round(value * GFLOP_PER_TILE.get(a.backbone, 0) / 1e3, 1),
            'step_issue': ((('three HIP graph replays (forward + backward to the cut, rest of the backward, clip + AdamW)' if replica.cut else 'two HIP graph replays (forward + backward, clip + AdamW)') + ' around the RCCL all-reduces of the gradient buckets (segland_amd/bucket_step.py, the train_base default at N > 1)' if replica is not None
                            else 'one HIP graph replay per step (segland_amd/graph_step.py, the train_base default on one GPU)') if graphed is not None
                           else 'kernel by kernel from Python' + (' under DistributedDataParallel / RCCL' if use_ddp else '')),
        }
        if eager_ms is not None:
            # the second region (the one the roofline events come from): kernel-by-kernel steps WITH event pairs around the roofline kernel
            out['roofline_region_ms_per_step'] = round(eager_ms, 3)
        per = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps))
        out['ms_per_step_median'] = round(per[len(per) // 2], 3)
        out['value_at_median'] = round(a.batch * world / (per[len(per) // 2] * 1e-3), 1)
        out['tflops_note'] = ('whole_step_tflops = tiles/s x the REFERENCE algorithm FLOPs per tile (%.1f GFLOP: what a plain implementation executes); '
                              'executed_tflops = tiles/s x the FLOPs this build launches on the MFMA kernels (collapsed head, pooled-grid PPM: see DESIGN.md 3)' % GFLOP_PER_TILE.get(a.backbone, 0))
        exec_gflop = sum(e['gflop'] for e in table.values())
        out['executed_tflops'] = round(value * exec_gflop / max(a.batch, 1) / 1e3, 1)
        fams = {}
        for k_, e in table.items():
            fams[k_] = {'tflops': round(e['gflop'] / max(e['ms_total'], 1e-9), 1), 'ms_per_step': round(e['ms_total'], 3), 'launches': e['calls'], 'floor_ms': round(e['floor_ms'], 3)}
            if e.get('gbytes'):
                # algorithmic operand + result bytes (activations in and out, weights once, residual addend + gate bits): the HBM-bound families
                # (conv_gemm_sk_kernel: K <= 256 1x1 convs; conv_c64k3_kernel) are judged on this figure against the 8 TB/s peak, the MFMA-bound ones on tflops
                fams[k_]['gb_per_s'] = round(e['gbytes'] / max(e['ms_total'], 1e-9) * 1e3, 0)
        for k_, e in table_bytes.items():
            fams[k_] = {'gb_per_s': round(e['gbytes'] / max(e['ms_total'], 1e-9) * 1e3, 0), 'ms_per_step': round(e['ms_total'], 3), 'launches': e['calls'], 'gbytes_per_step': round(e['gbytes'], 3),
                        'floor_ms': round(e['floor_ms'], 3)}
        out['families'] = fams               # one instrumented (un-timed) step: MFMA families in TFLOP/s, BatchNorm passes in algorithmic GB/s vs the 8 TB/s HBM peak
        # ---- the step's own speed of light: every instrumented launch priced at max(FLOP / dense MFMA peak, algorithmic bytes / 6.3 TB/s achievable HBM);
        # what is not instrumented (finalize / column-sum launches, pyramid row GEMMs, head combine, dispatch gaps) enters at its MEASURED time, so the floor errs high
        opt_key = 'optimizer (clip_grad_norm_ + AdamW)'
        inst_ms = sum(e['ms_total'] for e in table.values()) + sum(e['ms_total'] for e in table_bytes.values())
        inst_floor = sum(e['floor_ms'] for e in table.values()) + sum(e['floor_ms'] for e in table_bytes.values())
        step_ms = 1e3 * dt_s / a.steps
        rest_ms = max(0.0, step_ms - inst_ms)
        if opt_key in table_bytes:
            out['optimizer_ms_per_step'] = round(table_bytes[opt_key]['ms_total'], 3)
            out['optimizer_note'] = ('clip_grad_norm_(5.0) coefficient + both AdamW steps of train_base.py:262-264 (one launch), HIP events in the instrumented step; it is INSIDE '
                                     'ms_per_step and value (fwd + bwd alone: ms_per_step - optimizer_ms_per_step)')
        step_floor = {'step_floor_ms': round(inst_floor + rest_ms, 3), 'frac_of_floor': round((inst_floor + rest_ms) / max(step_ms, 1e-9), 4),
                      'instrumented_ms': round(inst_ms, 3), 'instrumented_floor_ms': round(inst_floor, 3), 'uninstrumented_ms_at_measured_time': round(rest_ms, 3),
                      'conv_floor_ms': round(sum(e['floor_ms'] for e in table.values()), 3), 'streaming_floor_ms': round(sum(e['floor_ms'] for e in table_bytes.values()), 3),
                      'note': 'sum over the launches of one instrumented step of max(FLOP / %g TFLOP/s, algorithmic bytes / 6.3 TB/s); families[*].floor_ms has it per kernel family; '
                              'frac_of_floor = step_floor_ms / ms_per_step (1.0 = the step runs at its own speed of light)' % peak}
        if live:
            e = list(live.values())[0]
            ach = e['gflop'] / max(e['ms_total'], 1e-9)        # GFLOP / ms == TFLOP/s
            # counter fields: read from the committed PMC passes of THIS library only (profiles/r5_*.json carry the sha256 of the libsegland_hip.so they were collected
            # with, tools/collect_traffic.py / tools/pmc_families.py); a different library -> null (a kernel change must not keep the old counters in the line)
            lib_sha = lib_sha16()
            traffic, tsrc = None, None
            for tname in ('r6_traffic.json', 'r6_traffic_wgrad.json', 'r6_traffic_p8.json', 'r5_traffic.json', 'r5_traffic_wgrad.json', 'r5_traffic_p8.json'):      # bytes per launch of one kernel each; the newest set whose sha matches
                tpath = os.path.join(ROOT, 'profiles', tname)
                if traffic is None and os.path.exists(tpath):
                    try:
                        t = json.load(open(tpath))
                        if t.get('kernel') and t['kernel'] == e['family']:
                            if t.get('lib_sha256_16') == lib_sha:
                                traffic = t.get('hbm_bytes_per_launch')
                                tsrc = ('NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command with this library (sha256 %s), committed as '
                                        'profiles/%s (tools/collect_traffic.py)' % (lib_sha, tname))
                            else:
                                tsrc = 'null: profiles/%s was collected with another build of the library (%s, loaded %s)' % (tname, t.get('lib_sha256_16'), lib_sha)
                    except Exception:
                        pass
            clock, mfma_busy, csrc = None, None, None
            cpath = os.path.join(ROOT, 'profiles', 'r6_pmc_families.json')
            if not os.path.exists(cpath):
                cpath = os.path.join(ROOT, 'profiles', 'r5_pmc_families.json')
            cname = os.path.basename(cpath)
            if os.path.exists(cpath) and a.model == 'pspnet_pop' and a.backbone == 'resnet50' and a.batch == 16 and a.size == 512 and a.dtype == 'bf16':      # the counter passes are of this workload
                try:
                    cj = json.load(open(cpath))
                    if cj.get('lib_sha256_16') != lib_sha:
                        csrc = 'null: profiles/%s was collected with another build of the library (%s, loaded %s)' % (cname, cj.get('lib_sha256_16'), lib_sha)
                    else:
                        base = e['family'].split('<')[0]
                        cands = [(v['ms_per_step'], v) for k_, v in cj.get('kernels', {}).items() if k_.split('<')[0].split('(')[0] == base]
                        if cands:
                            v = max(cands, key=lambda t: t[0])[1]
                            clock, mfma_busy = v.get('clock_ghz'), v.get('mfma_busy_over_sq_busy')
                            csrc = ('NOT measured in this run: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass of this command (kernel by kernel) with this '
                                    'library (sha256 %s), committed as profiles/%s (tools/collect_profiles.sh pmc): clock = GRBM_GUI_ACTIVE / kernel duration, '
                                    'mfma_busy = MFMA_BUSY / (32 x SQ_BUSY_CYCLES)' % (lib_sha, cname))
                except Exception:
                    pass
            # what north_star asks: MFMA utilisation "in the backbone convs" -- ALL conv / GEMM launches of the instrumented step, not the best kernel
            conv_gflop = sum(t_['gflop'] for t_ in table.values())
            conv_ms = sum(t_['ms_total'] for t_ in table.values())
            backbone_convs = {'gflop_executed_per_step': round(conv_gflop, 1), 'ms_per_step': round(conv_ms, 3), 'launches': sum(t_['calls'] for t_ in table.values()),
                              'tflops': round(conv_gflop / max(conv_ms, 1e-9), 1), 'frac': round(conv_gflop / max(conv_ms, 1e-9) / peak, 4),
                              'note': 'every conv / GEMM launch of one instrumented kernel-by-kernel step (forward, data and weight gradients of all layers incl. slab reduces): '
                                      'executed FLOPs / HIP-event time against the %g TFLOP/s peak; north_star target 0.60' % peak}
            out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                               'step_floor_ms': step_floor['step_floor_ms'], 'frac_of_floor': step_floor['frac_of_floor'], 'step_floor': step_floor,
                               'traffic': traffic, 'traffic_source': tsrc, 'clock_ghz_measured': clock, 'mfma_busy_measured': mfma_busy, 'clock_source': csrc,
                               'backbone_convs': backbone_convs, 'whole_step_frac': round(out['whole_step_tflops'] / peak, 4),
                               'kernel': e['family'], 'launches': e['calls'],
                               'ms_per_launch': round(e['ms_total'] / max(e['calls'], 1), 4),
                               'gflop_per_launch': round(e['gflop'] / max(e['calls'], 1), 2),
                               'note': 'all launches of this kernel in the %d instrumented kernel-by-kernel steps run right after the %d timed ones' % (a.steps, a.steps)
                                       + ' (every shape it serves; a weight-gradient span includes the small fixed-order slab reduce launched behind the kernel); achieved = sum of algorithmic FLOPs / sum of HIP-event time; '
                                       + 'peak is the 2.4 GHz figure: clock_ghz_measured is what the chip sustained under this kernel on N(0,1) operands in the counter pass (the same binary on '
                                       + 'all-zero operands runs +26...+36 % faster, profiles/r2_dvfs_zero_operands.txt, DESIGN.md 3.1c)'}
        if world == 1 and not use_ddp and not a.no_other_configs and a.model == 'pspnet_pop' and a.backbone == 'resnet50' and a.dtype == 'bf16' and a.batch == 16 and a.size == 512:
            # the other BASELINE configurations that fit one GPU, ~5 s each, each in a FRESH child process (round-4 advisor: a memory fault, hang or OOM kill in a side
            # measurement must not cost the headline line; the children are started with subprocess -- this GPU-initialised process never execs)
            model = opt = net = params = batches = graphed = eager_fn = fn = img = mask = replica = None      # noqa: F841
            import gc
            import subprocess
            gc.collect(); torch.cuda.empty_cache()
            others = {}
            for kind, key in (('c3', 'config3_resnet101_shard'), ('c4', 'config4_ft_pair'), ('c5', 'config5_swin_t_shard')):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--side-config', kind], capture_output=True, text=True, timeout=600,
                                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
                    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
                    others[key] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {'error': 'child exited with code %d: %s' % (r.returncode, (r.stderr or '').strip().splitlines()[-1:] or '')}
                except Exception as e:                  # noqa: BLE001
                    others[key] = {'error': '%s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else '')}
            out['other_configs'] = others
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(a.cpu_budget, a.backbone, a.model)
        print(json.dumps(out), flush=True)
    if use_ddp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
