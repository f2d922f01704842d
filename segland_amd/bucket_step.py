"""Data parallelism that a HIP graph can hold: gradient buckets owned by the build instead of DistributedDataParallel's reducer.

DistributedDataParallel runs its reducer (Python/C++ hooks inside backward) and RCCL work eagerly, so a step under it is ~750 (PSPNet-POP) /
~850 (Swin-POP) kernel launches issued from Python on every rank -- at the 8 tiles per GPU of the Swin configuration that is as long as the
GPU time itself.  `BucketedReplica` keeps the reference's semantics (engine.py:69-74: one replica per GPU, gradients
averaged over the ranks, `module.`-prefixed state_dict) with a structure that needs the host three times per step:

    graph A   zero_grad, forward, loss, backward -- the block backwards write every parameter gradient straight into flat fp32 buckets
              (functional.grad_dst; the few gradients autograd produces itself are copied in at the end of the graph)
    eager     one SUM all-reduce per bucket on the process group (RCCL over xGMI; nothing of RCCL is captured)
    graph B   gradient-norm clip + both AdamW steps of the loop body; the 1 / world_size of the mean lives in the optimizer kernel

The parameter-gradient exchange is the path's only collective.  Per-GPU BatchNorm statistics only: SyncBatchNorm's per-layer
collectives (SEGLAND_SYNC_BN=1) cannot sit inside a captured forward, `eligible()` says no and the caller keeps DistributedDataParallel.
What the two-graph form gives up is the overlap of the all-reduce with the backward (190 MB for ResNet-50: ~1 ms of a 26 ms step at 8 GPUs
against the 2.5-3 % the DistributedDataParallel wrapper costs at any world size); SEGLAND_BUCKET_STEP=0 keeps
DistributedDataParallel with the in-place bucket gradients of engine.enable_inplace_bucket_gradients."""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import graph_step


def eligible(world_size, use_cuda):
    """Bucketed replicas instead of DistributedDataParallel: GPU, per-GPU BatchNorm statistics, not switched off."""
    from . import functional
    return (use_cuda and os.environ.get('SEGLAND_BUCKET_STEP', '1') != '0' and os.environ.get('SEGLAND_STEP_GRAPH', '1') != '0'
            and functional._SYNC_BN == '0')


class BucketedReplica(nn.Module):
    """One model replica of a data-parallel job.  `.module` and the `module.` key prefix like DistributedDataParallel / nn.DataParallel.
    Construction broadcasts rank 0's parameters and buffers (DistributedDataParallel does the same); flat gradient buckets are laid out
    in REVERSE parameter order (the order the backward produces them), `cap_mb` each."""

    def __init__(self, module, process_group=None, cap_mb=64):
        super().__init__()
        self.module = module
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        if self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, 0, group=process_group)
        self.cap = int(cap_mb * (1 << 20)) // 4
        self.buckets, self.views, self.layout = [], {}, None
        self._build()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def _trainable(self):
        return [p for p in self.module.parameters() if p.requires_grad]

    def _build(self):
        params = self._trainable()
        layout = tuple((id(p), p.numel()) for p in params)
        if layout == self.layout:
            return
        for p in self.module.parameters():
            if hasattr(p, '_sl_gview'):
                del p._sl_gview
        self.buckets, self.views, chunk, n = [], {}, [], 0
        dev = params[0].device

        def flush():
            if chunk:
                flat = torch.zeros(sum(-(-p.numel() // 64) * 64 for p in chunk), dtype=torch.float32, device=dev)     # 256-byte aligned views
                off = 0
                for p in chunk:
                    v = flat[off:off + p.numel()].view(p.shape)
                    self.views[id(p)] = v
                    p._sl_gview = v
                    off += -(-p.numel() // 64) * 64
                self.buckets.append(flat)
        for p in reversed(params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('BucketedReplica: contiguous float32 parameters only')
            if n + p.numel() > self.cap and chunk:
                flush()
                chunk, n = [], 0
            chunk.append(p)
            n += p.numel()
        flush()
        self.layout = layout

    def adopt_gradients(self):
        """After backward: every .grad IS (an alias of) its bucket view.  Block backwards wrote most of them in place (functional.grad_dst);
        a gradient autograd produced itself is copied in; a parameter that received none counts as zero (DistributedDataParallel's
        find_unused_parameters=False would raise -- every parameter of the POP path receives a gradient)."""
        dst, src, zero = [], [], []
        for p in self._trainable():
            v = self.views[id(p)]
            g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr():
                dst.append(v); src.append(g)
            else:
                continue
            p.grad = v.detach()
        if dst:
            torch._foreach_copy_(dst, src)                # one multi-tensor launch (Swin-POP: ~170 gradients come from autograd itself)
        if zero:
            torch._foreach_zero_(zero)

    def all_reduce(self, which=None):
        """SUM over the ranks, one collective per bucket, on the process group's stream (ordered after the current stream's work)."""
        if self.active:                                   # also at world size 1 (SEGLAND_FORCE_DDP=1): the same calls, RCCL copies
            for k, flat in enumerate(self.buckets):
                if which is None or k in which:
                    dist.all_reduce(flat, group=self.group)

    def train_step_parts(self, optimizer, double_step=True, clip_grad=5.0):
        """(backward_part(img, mask) -> loss dict, update_part() -> gradient norm): train_base.train_iteration cut at the all-reduce."""
        from .optim import clip_coefficient

        def backward_part(img, mask):
            self._build()
            optimizer.zero_grad(set_to_none=True)
            loss_dict = self.module(img, mask)
            loss_dict['total_loss'].backward()
            self.adopt_gradients()
            return loss_dict

        def update_part():
            params = [p for p in self._trainable() if p.grad is not None]
            norm, coef = clip_coefficient(params, clip_grad, self.world)
            optimizer.step(repeat=2 if double_step else 1, grad_scale=coef)
            return norm
        return backward_part, update_part

    def train_iteration(self, optimizer, img, mask, double_step=True):
        """The loop body of train_base.py:250-264 issued kernel by kernel (what GraphedBucketStep replays)."""
        bwd, upd = self.train_step_parts(optimizer, double_step)
        loss = bwd(img, mask)
        self.all_reduce()
        return loss, upd()


class GraphedBucketStep:
    """Callable with the signature and results of train_base.train_iteration for a BucketedReplica: graph A, bucket all-reduces, graph B."""

    def __init__(self, replica, optimizer, double_step=True, warmup=3):
        bwd, upd = replica.train_step_parts(optimizer, double_step)
        self.replica = replica
        self.a = graph_step.GraphedStep(bwd, replica, None, warmup)
        self.b = graph_step.GraphedStep(lambda: upd(), replica, optimizer, warmup)

    def __call__(self, img, mask):
        loss = self.a(img, mask)
        self.replica.all_reduce()
        norm = self.b()
        return loss, norm

    @property
    def graph(self):
        return self.a.graph if (self.a.graph is not None and self.b.graph is not None) else None

    @property
    def replays(self):
        return min(self.a.replays, self.b.replays)
