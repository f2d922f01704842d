"""CPU tests of the host-side harness: command-line surface, LR schedule, parameter groups, checkpoint key format,
and the N > 1 data-parallel path over gloo (world_size 2) with the CPU oracle standing in for the HIP model
(the collectives, sharding and reductions of engine.py are device-agnostic)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden
from oracle import formula as fm
from oracle import pop_oracle as po

REF_BASE_FLAGS = ['--dataset', '--batch-size', '--data-dir', '--train-list', '--base-size', '--input-size', '--learning-rate', '--momentum',
                  '--power', '--weight-decay', '--start-epoch', '--num-epoch', '--random-seed', '--restore-from', '--snapshot-dir', '--model',
                  '--num-workers', '--backbone', '--os', '--print-frequency', '--save-pred-every', '--fold', '--shot', '--val-list',
                  '--test-batch-size', '--fix-bn', '--filter-novel', '--freeze-backbone', '--fp16', '--finetune']       # train_base.py:47-111
REF_FT_FLAGS = [f for f in REF_BASE_FLAGS if f != '--finetune'] + ['--update-base', '--update-epoch', '--fix-lr']       # ft_pop.py:47-115


def flags_of(parser):
    return {s for a in parser._actions for s in a.option_strings}


def test_cli_surface_matches_reference():
    from segland_amd.drivers import build_parser
    base, ft = build_parser(False), build_parser(True)
    assert set(REF_BASE_FLAGS) <= flags_of(base)
    assert set(REF_FT_FLAGS) <= flags_of(ft)
    a = base.parse_args([])
    assert (a.batch_size, a.learning_rate, a.weight_decay, a.power, a.num_epoch, a.random_seed, a.os, a.backbone) == (8, 1e-2, 0.0005, 0.9, 100, 321, 8, 'resnet50')
    f = ft.parse_args([])
    assert f.random_seed == '123,234' and f.fix_bn is True and f.update_epoch == 1
    # Engine-injected flags (engine.py:58-67 of the reference)
    from segland_amd.engine import Engine
    e = Engine(custom_parser=build_parser(False), argv=['--model', 'pspnet_pop', '--local_rank', '0'])
    assert {'-d', '--devices', '-c', '--continue', '--local_rank'} <= flags_of(e.parser) and not e.distributed


def test_lr_schedule_and_param_groups():
    from segland_amd.drivers import adjust_learning_rate_poly, lr_poly
    from segland_amd.networks.pspnet_pop import GFSS_Model
    from segland_amd.utils.pyt_utils import get_parameters
    assert lr_poly(1e-2, 0, 100, 0.9) == 1e-2
    np.testing.assert_allclose(lr_poly(1e-2, 50, 100, 0.9), 1e-2 * 0.5 ** 0.9)
    m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8)
    groups = get_parameters(m, lr=1e-3)
    sizes = [(len(g['params']), sum(p.numel() for p in g['params'])) for g in groups]
    assert sizes == [(159, 23508032), (6, 3072), (15, 23861760)]                     # SURVEY 8 a-12, == golden g8
    assert np.array_equal(np.array(sizes), golden('g8_traj')['group_sizes'])
    assert groups[1]['weight_decay'] == 0.0 and groups[1]['lr'] == 1e-2 and groups[0]['lr'] == 1e-3
    opt = torch.optim.AdamW(groups, lr=1e-3)
    lr = adjust_learning_rate_poly(opt, 1e-3, 10, 100, 0.9, split=0)
    assert opt.param_groups[0]['lr'] == lr and opt.param_groups[2]['lr'] == lr * 10
    # ft model: only novel_emb + classifier_n are trainable (SURVEY 3.2 [probe])
    mf = GFSS_Model(n_base=7, is_ft=True, n_novel=4, backbone='resnet50', pretrained_model=None, dilated=True, os=8)
    assert sorted(k for k, p in mf.named_parameters() if p.requires_grad) == ['classifier_n.0.weight', 'classifier_n.2.weight', 'classifier_n.4.weight', 'novel_emb']


def test_checkpoint_format_roundtrip(tmp_path):
    from segland_amd.drivers import save_checkpoint
    from segland_amd.engine import ModuleWrapper
    from segland_amd.networks.pspnet_pop import GFSS_Model
    from segland_amd.utils.pyt_utils import load_model
    m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8)
    fm.load_formula_weights(m)
    path = str(tmp_path / 'epoch_1.pth')
    save_checkpoint(ModuleWrapper(m), path)
    sd = torch.load(path, map_location='cpu')
    assert all(k.startswith('module.') for k in sd) and 'module.decoder.bottleneck.3.bias' in sd       # the reference's key format
    # ... which loads back both into our model and into the oracle (== reference layout) with is_restore=True
    m2 = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8)
    load_model(m2, path, is_restore=True)
    assert torch.equal(m2.backbone.layer3[4].conv2.weight, m.backbone.layer3[4].conv2.weight)
    o = po.PopOracle(7)
    o.load_state_dict({k[7:]: v for k, v in sd.items()}, strict=True)


def test_torchvision_keyed_resnet_loads_as_pretrained_backbone(tmp_path):
    """Row f-4: an ImageNet ResNet-50 checkpoint in torchvision's key layout (conv1 / bn1 / layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} /
    downsample.{0,1} + the classification head fc.*) through get_backbone(pretrained_model=...) -> utils.pyt_utils.load_model
    (networks/backbones/__init__.py:41-42, utils/pyt_utils.py:86-135): every backbone tensor is taken, fc.* is reported as unexpected and dropped."""
    import logging
    from segland_amd.networks.backbones import get_backbone
    from segland_amd.networks.pspnet_pop import GFSS_Model
    torch.manual_seed(7)
    # torchvision.models.resnet50().state_dict(): same names and shapes as our backbone (stride-2 convs live in conv2 there too) plus fc
    donor = get_backbone(norm_layer=torch.nn.BatchNorm2d, backbone='resnet50', dilated=False, os=32)
    sd = {k: torch.randn_like(v) if v.dtype.is_floating_point else v.clone() for k, v in donor.state_dict().items()}
    assert len(sd) == 318 and 'layer4.2.bn3.running_var' in sd and 'layer1.0.downsample.0.weight' in sd         # 320 torchvision entries minus fc.*
    sd['fc.weight'], sd['fc.bias'] = torch.randn(1000, 2048), torch.randn(1000)
    path = str(tmp_path / 'resnet50-imagenet.pth')
    torch.save(sd, path, _use_new_zipfile_serialization=False)
    seen = []
    h = logging.Handler(); h.emit = lambda r: seen.append(r.getMessage())
    logging.getLogger().addHandler(h)
    try:
        m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=path, dilated=True, os=8)
    finally:
        logging.getLogger().removeHandler(h)
    for k, v in m.backbone.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert any('Unexpected key' in s and 'fc.weight' in s for s in seen) and not any('Missing key' in s for s in seen)


def test_gpu_only_model_fails_loudly_on_cpu():
    from segland_amd.networks.pspnet_pop import GFSS_Model
    m = GFSS_Model(n_base=7, backbone='resnet50', pretrained_model=None, dilated=True, os=8)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(2, 3, 64, 64))


# ------------------------------------------------------------------------------------------------ world_size 2 over gloo
def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _small_oracle():
    """A 2-block 'backbone' oracle: same code path as the full model (stem, bottlenecks, PPM, POP head, OrthLoss), sized for CPU."""
    torch.manual_seed(0)
    bb = po._Holder()
    bb.conv1 = po._conv(3, 64, 7, stride=2, pad=3); bb.bn1 = torch.nn.BatchNorm2d(64)
    bb.layer1 = torch.nn.Sequential(po.make_bottleneck(64, 16, 1, 1, True)); bb.layer2 = torch.nn.Sequential(po.make_bottleneck(64, 16, 2, 1, True))
    bb.layer3 = torch.nn.Sequential(); bb.layer4 = torch.nn.Sequential()
    m = po.PopOracle(n_base=7, criterion=po.OrthLossOracle(255), _custom_backbone=bb, feat_channels=64)
    fm.load_formula_weights(m)
    return m


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from segland_amd.drivers import build_parser
    from segland_amd.engine import Engine
    with Engine(custom_parser=build_parser(False), argv=['--model', 'pspnet_pop', '--batch-size', '4']) as engine:
        assert engine.distributed and engine.world_size == world and dist.get_backend() == 'gloo'
        m = _small_oracle()
        po.train_mode(m)                          # BN in eval: DP over a split batch == one process over the full batch
        model = engine.data_parallel(m)
        img = fm.formula_image(4, 64, 64, 'ddp/img'); mask = fm.formula_mask(4, 64, 64, 8, 'ddp/mask', block=16, ignore_rows=0)
        sl = slice(rank * 2, rank * 2 + 2)        # the sampler's shard: per-rank batch = global / world (engine.py:84-86)
        d = model(img[sl], mask[sl])
        d['total_loss'].backward()
        vals = engine.reduce_loss_dict(d)
        inter = engine.all_reduce_tensor(torch.tensor([1.0 + rank, 2.0]), norm=False)
        grads = {k: p.grad.clone() for k, p in model.module.named_parameters() if p.grad is not None}
        if rank == 0:
            q.put((vals, inter.tolist(), {k: v.numpy() for k, v in list(grads.items())[:6]}, float(torch.stack([g.norm() for g in grads.values()]).norm())))


def test_ddp_over_gloo_matches_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    vals, inter, grads, gnorm = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process on the full batch.  DDP averages gradients over ranks; the mean-over-valid-pixels CE of the full
    # batch equals the mean of the two shard losses here because both shards have the same number of valid pixels.
    m = _small_oracle(); po.train_mode(m)
    img = fm.formula_image(4, 64, 64, 'ddp/img'); mask = fm.formula_mask(4, 64, 64, 8, 'ddp/mask', block=16, ignore_rows=0)
    d = m(img, mask); d['total_loss'].backward()
    np.testing.assert_allclose(vals['total_loss'], float(d['total_loss']), rtol=1e-5)
    np.testing.assert_allclose(vals['seg_loss'], float(d['seg_loss']), rtol=1e-5)
    assert inter == [3.0, 4.0]
    ref = dict(m.named_parameters())
    for k, g in grads.items():
        np.testing.assert_allclose(g, ref[k].grad.numpy(), rtol=2e-4, atol=1e-7, err_msg=k)
    ref_norm = float(torch.stack([p.grad.norm() for p in m.parameters() if p.grad is not None]).norm())
    np.testing.assert_allclose(gnorm, ref_norm, rtol=1e-4)


def test_bench_self_launch_starts_fresh_child_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher (how the driver may invoke the scaling bench): the parent starts N ranks through
    torch.distributed.run as CHILD processes before touching the GPU; on a node with fewer GPUs it reports the rank-count problem."""
    import importlib.util
    import subprocess
    import sys
    import torch
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, 'visible_gpus', lambda: 1)
    assert bench.self_launch(2, ['--gpus', '2']) == 3
    assert 'exposes 1 GPU' in capsys.readouterr().err
    calls = []
    monkeypatch.setattr(bench, 'visible_gpus', lambda: 8)
    monkeypatch.setattr(subprocess, 'call', lambda cmd, env=None: calls.append((cmd, env)) or 0)
    assert bench.self_launch(4, ['--gpus', '4', '--steps', '5']) == 0
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert '--master-addr' in cmd and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '4', '--steps', '5']
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # end to end in a real child: no GPU in this container -> the rank-count message and exit code 3, not a usage error
    r = subprocess.run([sys.executable, bench.__file__, '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
    assert r.returncode == 3 and 'one rank per GPU' in r.stderr and 'usage' not in r.stderr.lower(), (r.returncode, r.stderr[-500:])


# ------------------------------------------------------------------------------------------------ bucket step over gloo (CPU): layout, backward cut, collectives
def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', init_method='env://')
    from segland_amd import bucket_step
    m = _small_oracle()
    po.train_mode(m)
    if rank == 1:
        with torch.no_grad():
            m.base_emb.add_(0.5)                              # the replica must take rank 0's parameters
    rep = bucket_step.BucketedReplica(m, cap_mb=0.05, cut=False)
    assert len(rep.buckets) >= 3 and rep.late_buckets == len(rep.buckets)
    img = fm.formula_image(4, 64, 64, 'ddp/img'); mask = fm.formula_mask(4, 64, 64, 8, 'ddp/mask', block=16, ignore_rows=0)
    sl = slice(rank * 2, rank * 2 + 2)
    for p in m.parameters():
        p.grad = None
    d = rep(img[sl], mask[sl])
    d['total_loss'].backward()
    rep.adopt_gradients()
    assert all(p.grad.data_ptr() == rep.views[id(p)].data_ptr() for p in m.parameters() if p.requires_grad)
    works = rep.all_reduce(async_op=True)
    for w in works:
        w.wait()
    grads = {k: p.grad.clone() / world for k, p in m.named_parameters() if p.grad is not None}
    if rank == 0:
        q.put({k: v.numpy() for k, v in grads.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_replica_over_gloo_matches_single_process():
    """bucket_step.BucketedReplica on two gloo ranks (CPU): parameters broadcast from rank 0, every gradient adopted into the flat buckets, one SUM all-reduce per
    bucket; sum / world_size equals the gradient of the one-process full batch (BatchNorm in eval, as in the DDP test above)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m = _small_oracle()
    po.train_mode(m)
    img = fm.formula_image(4, 64, 64, 'ddp/img'); mask = fm.formula_mask(4, 64, 64, 8, 'ddp/mask', block=16, ignore_rows=0)
    d = m(img, mask)
    d['total_loss'].backward()
    for k, p in m.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(got[k], p.grad.numpy(), rtol=2e-4, atol=1e-6, err_msg=k)


def test_bucket_replica_backward_cut_on_cpu():
    """The two-part backward of bucket_step.BucketedReplica on a toy model with the cut interface of the GPU models (late_parameters / cut_tensors on a detached
    leaf / clear_cut): gradients of both halves land in their own bucket groups and equal a plain backward."""
    from segland_amd import bucket_step

    Toy = _CutToy
    x, y = torch.randn(5, 8), torch.randn(5, 4)
    ref = Toy()
    ref(x, y)['total_loss'].backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    m = Toy()
    rep = bucket_step.BucketedReplica(m, cap_mb=0.001, cut=True)          # (the default cuts only at world size > 1)
    assert rep.cut and m._want and 0 < rep.late_buckets < len(rep.buckets)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    bwd1, bwd2, _ = rep.train_step_parts(opt)
    bwd1(x, y)
    assert all(p.grad is not None for p in m.late.parameters()) and all(p.grad is None for p in m.early.parameters())
    bwd2()
    assert m._cut is None
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, want[k], rtol=1e-6, atol=1e-7), k
        assert p.grad.data_ptr() == rep.views[id(p)].data_ptr()
    late_ids = {id(p) for p in m.late.parameters()}
    n_late = sum(-(-p.numel() // 64) * 64 for p in m.parameters() if id(p) in late_ids)
    assert sum(b.numel() for b in rep.buckets[:rep.late_buckets]) == n_late


# ------------------------------------------------------------------------------------------------ the capture decision of GraphedBucketStep is collective
class _CutToy(torch.nn.Module):
    """The cut interface of the GPU models (late_parameters / cut_tensors on a detached leaf / clear_cut) on two small MLP halves."""
    bucket_cut_default = True

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.early = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16))
        self.late = torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(16, 4))
        self._want, self._cut = False, None

    def enable_backward_cut(self, flag):
        self._want = bool(flag)

    def late_parameters(self):
        return list(self.late.parameters())

    def cut_tensors(self):
        return [self._cut] if self._cut is not None else None

    def clear_cut(self):
        self._cut = None

    def forward(self, x, y):
        self._cut = None
        h = self.early(x)
        if self._want and h.requires_grad:
            leaf = h.detach().requires_grad_(True)
            self._cut, h = (h, leaf), leaf
        return {'total_loss': ((self.late(h) - y) ** 2).mean()}


class _ToyOpt(torch.optim.SGD):
    """segland_amd.optim.AdamW's extra surface (step(repeat, grad_scale), capture_begin, graph_prepare) on plain SGD."""

    def step(self, closure=None, repeat=1, grad_scale=None):
        for _ in range(repeat):
            for g in self.param_groups:
                for p in g['params']:
                    if p.grad is not None:
                        p.data.add_(p.grad * (grad_scale if grad_scale is not None else 1.0), alpha=-g['lr'])

    def capture_begin(self):
        pass

    def graph_prepare(self):
        pass


def _fallback_worker(rank, world, port, q, fail_rank):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', init_method='env://')
    from segland_amd import bucket_step

    class Recorded:
        """Stands in for a captured HIP graph on the CPU: recording does not run the part, replay() does."""

        def __init__(self, fn, args, slot):
            self.fn, self.args, self.slot = fn, args, slot

        def replay(self):
            out = self.fn(*self.args)
            if isinstance(out, dict):
                self.slot.update({k: v.detach() for k, v in out.items()})
            elif out is not None:
                self.slot['norm'] = out.detach()

    class Step(bucket_step.GraphedBucketStep):
        attempts = 0

        def _capture_graphs(self, img, mask):
            Step.attempts += 1
            graphs, loss, norm = [], {}, {}
            self.static_in = (img.clone(), mask.clone())

            def call(fn, *args):
                if rank == fail_rank and Step.attempts == 1 and len(graphs) == 1:
                    raise RuntimeError('injected: the second part fails to capture on rank %d only' % rank)
                graphs.append(Recorded(fn, self.static_in if args else (), loss if not graphs else norm))
                return loss if len(graphs) == 1 else norm
            self.replica._run(self.parts, img, mask, call, collectives=False)
            return graphs, (loss, norm)

    torch.manual_seed(100 + rank)
    data = [(torch.randn(5, 8), torch.randn(5, 4)) for _ in range(5)]

    def run(graphed):
        m = _CutToy()
        rep = bucket_step.BucketedReplica(m, cap_mb=0.001)
        opt = _ToyOpt(m.parameters(), lr=0.05)
        step = Step(rep, opt, double_step=True, warmup=1) if graphed else None
        for x, y in data:
            step(x, y) if graphed else rep.train_iteration(opt, x, y, double_step=True)
        return [p.detach().clone() for p in m.parameters()], step

    want, _ = run(False)
    got, step = run(True)
    same = all(torch.equal(a, b) for a, b in zip(want, got))
    # the ranks hold the same parameters after every step (same averaged gradient): compare rank 1's with rank 0's
    flat = torch.cat([p.reshape(-1) for p in got])
    ref0 = flat.clone()
    dist.broadcast(ref0, 0)
    q.put((rank, same, bool(torch.equal(flat, ref0)), step.failures, step.replays, Step.attempts))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('fail_rank', [1, 0])
def test_bucket_step_capture_failure_on_one_rank_is_a_collective_decision(fail_rank):
    """Round-3 advisor: a capture that fails on ONE rank (out of memory on one GPU) must not leave that rank issuing the step's all-reduces twice while the other issues
    them once.  GraphedBucketStep captures without running or exchanging anything, the ranks agree, and the step runs once everywhere: kernel by kernel after a failed
    attempt (on either rank), as three replays after a good one.  Two gloo ranks, recorded parts standing in for HIP graphs; parameters equal the plain
    train_iteration run bit for bit and equal across ranks; both ranks count one failure and the same number of replays (no hang = the collectives lined up)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same, same_as_rank0, failures, replays, attempts in res:
        assert same, 'rank %d: parameters differ from the kernel-by-kernel run' % rank
        assert same_as_rank0, 'rank %d: parameters differ from rank 0' % rank
        assert (failures, attempts, replays) == (1, 2, 3), (rank, failures, attempts, replays)      # step 0 warm-up, 1 failed attempt -> eager, 2 capture + replay, 3-4 replays


def _agree_worker(rank, world, port, q, case):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', init_method='env://')
    from segland_amd import bucket_step
    rep = bucket_step.BucketedReplica(_CutToy(), cap_mb=0.001)
    import time
    if case == 'signatures':
        # every rank says its capture went through, but rank 1 captured another input shape: nobody may replay
        got = rep.agree(True, signature=((5, 8), 'f32') if rank == 0 else ((4, 8), 'f32'), timeout_s=60)
        same = rep.agree(True, signature=((5, 8), 'f32'), timeout_s=60)                       # ... and the group still works afterwards
        q.put((rank, got, same, None))
        dist.barrier()
    else:
        # rank 1 never reaches the handshake (its step sequence diverged): rank 0 must come back with an error inside the bound instead of hanging
        if rank == 0:
            t0, err = time.time(), None
            try:
                rep.agree(True, signature='x', timeout_s=3.0)
            except RuntimeError as e:
                err = str(e)
            q.put((rank, time.time() - t0, None, err))
        else:
            time.sleep(8.0)
            q.put((rank, 0.0, None, 'skipped'))
    try:
        dist.destroy_process_group()
    except Exception:                # noqa: BLE001  (the timed-out handshake may have left the group unusable: the process is about to end)
        pass


@pytest.mark.parametrize('case', ['signatures', 'skipped'])
def test_bucket_step_agree_mismatched_signatures_and_bounded_wait(case):
    """Round-5 advisor: the two untested branches of BucketedReplica.agree on two gloo ranks.  'signatures': both ranks report a successful capture of DIFFERENT input
    signatures -> agree() is False on both (all stay kernel by kernel) and the next handshake works.  'skipped': one rank never calls agree -> the other gets a
    RuntimeError naming the cause within the bound (3 s here), not a hang."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    if case == 'signatures':
        for rank, got, same, _ in res:
            assert got is False and same is True, (rank, got, same)
        assert all(p.exitcode == 0 for p in procs)
    else:
        rank, took, _, err = res[0]
        assert err is not None and 'did not all reach a capture attempt' in err, err
        assert took < 30.0, took


def test_prediction_tiff_writer(tmp_path):
    """eval_base.py:180-188 (the label map of a tile as a single-band uint8 TIFF with the class colormap): without rasterio (this image) Pillow writes a palette TIFF with
    the same pixels and the same colours; read back through the product's own tile decoder and through Pillow."""
    import numpy as np
    from segland_amd.eval_base import write_prediction_tiff
    from segland_amd.fusemat import COLORMAP
    rng = np.random.default_rng(3)
    pred = rng.integers(0, 12, size=(96, 130), dtype=np.uint8)
    path = str(tmp_path / 'tile.tif')
    how = write_prediction_tiff(path, pred, source_tif=None)
    assert how in ('PIL', 'rasterio')
    from PIL import Image
    img = Image.open(path)
    assert img.mode == 'P' and img.size == (130, 96)
    assert np.array_equal(np.asarray(img), pred)
    pal = np.asarray(img.getpalette()[:3 * 12], dtype=np.uint8).reshape(12, 3)
    assert np.array_equal(pal, np.resize(COLORMAP, (256, 3))[:12].astype(np.uint8))
