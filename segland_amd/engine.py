"""Process-group runtime of the drivers (counterpart of the reference's engine.py:23-141).

One process per GPU.  On MI355X the backend string "nccl" resolves to RCCL and the gradient all-reduce travels over
xGMI; on a CPU-only host (unit tests) the same code runs over gloo.  Differences from the reference, on purpose:
  * one process per GPU also on a single GPU (no nn.DataParallel); a pass-through wrapper keeps the `module.` key prefix
    of the reference's checkpoints and the `model.module.<attr>` access pattern of its drivers;
  * DDP is built with find_unused_parameters=False (every parameter of the POP path receives a gradient; the reference's
    True forces a graph walk per iteration), gradient_as_bucket_view=True and broadcast_buffers=False (BatchNorm
    statistics are per GPU -- see DESIGN.md "Multi-GPU");
  * loss scalars are reduced as ONE 3-float all-reduce, and only when they are printed (the reference does three
    blocking all-reduce + .item() round trips per iteration, train_base.py:266-267).
"""
import argparse
import os

import torch
import torch.distributed as dist
import torch.nn as nn

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL on this driver stack


class ModuleWrapper(nn.Module):
    """Single-process stand-in for nn.DataParallel: same `.module` attribute and `module.`-prefixed state_dict."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _allreduce_sum_hook(state, bucket):
    """DDP communication hook: all-reduce (SUM) of the flat bucket, no division (the optimizer kernel applies 1 / world_size)."""
    fut = dist.all_reduce(bucket.buffer(), group=state, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])


def _cache_bucket_views(optimizer, args, kwargs):
    # before every optimizer step the gradients ARE DDP's bucket views (gradient_as_bucket_view): remember them for the next backward
    for group in optimizer.param_groups:
        for p in group['params']:
            if p.grad is not None:
                p._sl_gview = p.grad


_hook_registered = []


def enable_inplace_bucket_gradients(ddp, process_group=None):
    """Sum-only all-reduce hook + bucket-view cache on `ddp` (see functional.grad_dst).  Gradients arrive SUMMED over the ranks."""
    ddp.register_comm_hook(process_group, _allreduce_sum_hook)
    if not _hook_registered:
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        _hook_registered.append(register_optimizer_step_pre_hook(_cache_bucket_views))
    return ddp


_FORKSERVER = []


def worker_context(num_workers, use_cuda):
    """How DataLoader workers are started next to a GPU process: from a FORK SERVER (a small helper process with torch and the dataset package imported, started once)
    instead of by forking the training process itself.  The loaders are re-created every epoch (like the reference's); fork() copies the page tables of everything the
    parent has mapped -- after a 28 GB host-side oracle run in the same process the driver tests spent 50-115 s starting workers that take 3 s in a fresh process
    (gpurun r4a / r4e), and a training process with pinned buffers and the HIP runtime's mappings pays the same per worker and epoch.  Measured: fork server
    start 0.8 s once, 0.02 s per loader afterwards (tools/loader_start.py).  Workers only decode and draw (nothing in them touches the GPU); datasets and collate
    objects are plain picklable classes.  CPU-only runs keep torch's default."""
    if num_workers <= 0 or not use_cuda:
        return None
    import multiprocessing as mp
    if not _FORKSERVER:
        try:
            mp.set_forkserver_preload(['torch', 'numpy', 'segland_amd.dataset'])
        except Exception:               # noqa: BLE001  (a server started by someone else keeps its own preload list)
            pass
        _FORKSERVER.append(mp.get_context('forkserver'))
    return _FORKSERVER[0]


class Engine(object):
    def __init__(self, custom_parser=None, argv=None):
        self.parser = custom_parser if custom_parser is not None else argparse.ArgumentParser()
        self.inject_default_parser()
        self.args = self.parser.parse_args(argv)
        self.continue_state_object = self.args.continue_fpath
        self.world_size = int(os.environ.get('WORLD_SIZE', '1'))
        # SEGLAND_FORCE_DDP=1: the DDP / RCCL code path also at world size 1 (tests and A/B timing on one GPU)
        self.distributed = self.world_size > 1 or os.environ.get('SEGLAND_FORCE_DDP') == '1'
        self.local_rank = int(os.environ.get('LOCAL_RANK', self.args.local_rank))
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            torch.cuda.set_device(self.local_rank)
        self.device = torch.device('cuda', self.local_rank) if self.use_cuda else torch.device('cpu')
        if self.distributed and not dist.is_initialized():
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29500')
            dist.init_process_group(backend='nccl' if self.use_cuda else 'gloo', init_method='env://')
        self.rank = dist.get_rank() if self.distributed else 0
        self.grad_div = 1
        self.devices = list(range(self.world_size))

    def inject_default_parser(self):
        p = self.parser
        p.add_argument('-d', '--devices', default='', help='set data parallel training')
        p.add_argument('-c', '--continue', type=str, metavar='FILE', dest='continue_fpath', help='continue from one certain checkpoint')
        p.add_argument('--local_rank', default=0, type=int, help='process rank on node')

    @property
    def is_main(self):
        return self.rank == 0

    def data_parallel(self, model, sum_gradients=False, graphable=False):
        """sum_gradients=True (segland_amd.optim.AdamW on the GPU): the all-reduce SUMS and the 1 / world_size of DDP's mean is folded into the
        optimizer kernel's gradient scale (`self.grad_div`, handed to NativeScalerWithGradNormCount) -- together with the in-place gradient
        writes of functional.grad_dst this removes DDP's per-parameter copy and scale kernels.  Otherwise: stock DDP averaging.
        graphable=True (train_base with the step graph on): a bucket_step.BucketedReplica instead of DistributedDataParallel when eligible."""
        model = model.to(self.device)
        self.grad_div = 1
        if not self.distributed:
            return ModuleWrapper(model)
        from . import bucket_step
        if sum_gradients and graphable and bucket_step.eligible(self.world_size, self.use_cuda):
            # gradient buckets owned by the build: the step is two HIP graphs around one all-reduce per bucket (bucket_step.py) instead of ~750
            # launches issued from Python under DistributedDataParallel's reducer
            self.grad_div = self.world_size
            return bucket_step.BucketedReplica(model, cap_mb=64)
        kw = dict(find_unused_parameters=False, gradient_as_bucket_view=True, broadcast_buffers=False, bucket_cap_mb=64)
        if self.use_cuda:
            ddp = nn.parallel.DistributedDataParallel(model, device_ids=[self.local_rank], output_device=self.local_rank, **kw)
        else:
            ddp = nn.parallel.DistributedDataParallel(model, **kw)
        if sum_gradients:
            enable_inplace_bucket_gradients(ddp)
            self.grad_div = self.world_size
        return ddp

    def _loader(self, dataset, batch_size, num_workers, train):
        sampler = None
        if self.distributed:
            sampler = torch.utils.data.distributed.DistributedSampler(dataset)
            batch_size = max(1, batch_size // self.world_size)          # engine.py:86 of the reference
            num_workers = num_workers // self.world_size if train else num_workers
        raw = getattr(dataset, 'raw_tiles', False)           # OpenEarthMap readers: workers decode, the GPU prepares the batch (dataset/augment.py)
        collate = None
        if raw:
            from .dataset.oem import raw_collate
            collate = getattr(dataset, 'collate_fn', None) or raw_collate        # fine-tune pair readers bring their own (dataset/oem_ft.py)
        loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, drop_last=train,
                                             shuffle=(train and sampler is None), pin_memory=self.use_cuda and not raw, sampler=sampler,
                                             collate_fn=collate, multiprocessing_context=worker_context(num_workers, self.use_cuda))
        return loader, sampler

    def get_train_loader(self, train_dataset):
        return self._loader(train_dataset, self.args.batch_size, self.args.num_workers, True)

    def get_test_loader(self, test_dataset):
        return self._loader(test_dataset, self.args.test_batch_size, self.args.num_workers, False)

    def all_reduce_tensor(self, tensor, norm=True):
        if not self.distributed:
            return tensor
        with torch.no_grad():
            t = tensor.detach().clone()
            dist.all_reduce(t, dist.ReduceOp.SUM)
            if norm:
                t.div_(self.world_size)
        return t

    def reduce_loss_dict(self, loss_dict):
        """{name: python float}, averaged over ranks, with ONE collective and ONE device->host copy."""
        keys = list(loss_dict.keys())
        vec = torch.stack([loss_dict[k].detach().float().reshape(()) for k in keys])
        vec = self.all_reduce_tensor(vec, norm=True)
        return dict(zip(keys, vec.tolist()))

    def __enter__(self):
        return self

    def __exit__(self, type, value, tb):
        if self.use_cuda:
            torch.cuda.empty_cache()
        if self.distributed and dist.is_initialized():
            dist.destroy_process_group()
        return False
