"""Data parallelism that a HIP graph can hold: gradient buckets owned by the build instead of DistributedDataParallel's reducer.

DistributedDataParallel runs its reducer (Python/C++ hooks inside backward) and RCCL work eagerly, so a step under it is ~750 (PSPNet-POP) /
~850 (Swin-POP) kernel launches issued from Python on every rank -- at the 8 tiles per GPU of the Swin configuration that is as long as the
GPU time itself.  `BucketedReplica` keeps the reference's semantics (engine.py:69-74: one replica per GPU, gradients
averaged over the ranks, `module.`-prefixed state_dict) with a structure that needs the host three times per step:

    graph A1  zero_grad, forward, loss, backward down to the model's cut tensors (ResNet: the output of layer3) -- the block backwards write every
              parameter gradient straight into flat fp32 buckets (functional.grad_dst; the few gradients autograd produces itself are copied in)
    eager     SUM all-reduce of the LATE buckets (head, decoder, layer4: 82 % of ResNet-50's bytes), asynchronous on the process group's stream
    graph A2  the rest of the backward (runs while those buckets travel)
    eager     SUM all-reduce of the early buckets; the current stream then waits for all of them (RCCL over xGMI; nothing of RCCL is captured)
    graph B   gradient-norm clip + both AdamW steps of the loop body; the 1 / world_size of the mean lives in the optimizer kernel

The parameter-gradient exchange is the path's only collective with per-GPU BatchNorm statistics (the default).  SyncBatchNorm's per-layer
collectives (SEGLAND_SYNC_BN=1: the reference's distributed semantics, train_base.py:175-178) cannot sit inside a captured forward: the replica is used all the
same -- same buckets, same in-place gradients, same collectives around the two backward halves -- but `GraphedBucketStep` then issues every step kernel by kernel
(`capturable()` says why) instead of handing the job to DistributedDataParallel's reducer.
Capturing is a COLLECTIVE decision: the three parts are captured back to back without running anything and without any collective, the ranks then agree (MIN
all-reduce of a success flag) and either all replay or all stay kernel by kernel -- a rank whose capture failed alone would otherwise issue the step's all-reduces
a second time while the others issue them once (round-3 advisor).
Exposed communication: the early buckets only (34 MB for ResNet-50 against 190 MB without the cut, SEGLAND_BUCKET_CUT=0); the DistributedDataParallel wrapper
costs 2-3 % at any world size.  SEGLAND_BUCKET_STEP=0 keeps DistributedDataParallel with the in-place bucket gradients of
engine.enable_inplace_bucket_gradients."""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import graph_step


def eligible(world_size, use_cuda):
    """Bucketed replicas instead of DistributedDataParallel: GPU, not switched off.  (With SEGLAND_SYNC_BN=1 the replica's step is issued kernel by kernel.)"""
    return use_cuda and os.environ.get('SEGLAND_BUCKET_STEP', '1') != '0' and os.environ.get('SEGLAND_STEP_GRAPH', '1') != '0'


def capturable(module):
    """None when the replica's step can be captured, else the reason it is issued kernel by kernel: a forward / backward that contains collectives of its own
    (SyncBatchNorm statistics, functional.sync_world) cannot be a HIP graph on this stack -- nothing of RCCL is captured."""
    from . import functional
    if functional._SYNC_BN != '0' and any(isinstance(m, nn.SyncBatchNorm) for m in module.modules()):
        return 'SEGLAND_SYNC_BN=%s: SyncBatchNorm all-reduces inside the forward and the backward' % functional._SYNC_BN
    return None


class BucketedReplica(nn.Module):
    """One model replica of a data-parallel job.  `.module` and the `module.` key prefix like DistributedDataParallel / nn.DataParallel.
    Construction broadcasts rank 0's parameters and buffers (DistributedDataParallel does the same); flat gradient buckets are laid out
    in REVERSE parameter order (the order the backward produces them), `cap_mb` each.

    cut (default on when the model offers it): the backward is run in two halves -- from the loss down to the model's cut tensors (ResNet: the output of
    layer3; Swin: the backbone's four feature maps), then from there to the image -- with the gradients of the LATE parameters (head, decoder, layer4: 82 % of
    ResNet-50's bytes) in buckets of their own: their all-reduce travels while the second half runs."""

    def __init__(self, module, process_group=None, cap_mb=64, cut=None):
        super().__init__()
        self.module = module
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        # the capture handshake (agree) runs on a group of its own: a rank whose step sequence diverged then waits THERE while the others' bucket all-reduces wait on
        # the data group -- both sides time out with a message, instead of a 3-float all-reduce pairing up with a bucket all-reduce of another size (round-5 advisor)
        self.hs_group = dist.new_group(ranks=dist.get_process_group_ranks(process_group) if process_group is not None else None) if (self.active and self.world > 1) else process_group
        if self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, 0, group=process_group)
        if cut is None:
            # The cut buys overlap of the late buckets' all-reduce with the second half of the backward and costs a third graph and a second round of collectives:
            # measured at world size 1 (profiles/r3_ddp_overhead*.txt) 0.4 ms per ResNet-50 step and 0.8 ms per Swin-T step with nothing to hide.  So: off at world
            # size 1; at N > 1 the model says which way it goes (ResNet: on -- 156 of 190 MB travel beside layer1-3's backward; Swin-T: off).  What it hides at 8 GPUs
            # is an ESTIMATE (~0.85 ms for ResNet-50 at ~300 GB/s of bus bandwidth): no multi-GPU node was available to measure it.  SEGLAND_BUCKET_CUT=1 / 0 overrides.
            env = os.environ.get('SEGLAND_BUCKET_CUT')
            cut = (env != '0') if env is not None else (bool(getattr(module, 'bucket_cut_default', False)) and self.world > 1)
        self.cut = bool(cut) and hasattr(module, 'late_parameters') and hasattr(module, 'cut_tensors')
        self.cap = int(cap_mb * (1 << 20)) // 4
        self.buckets, self.late_buckets, self.views, self.layout = [], 0, {}, None
        self._cuts = None
        self._build()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def _trainable(self):
        return [p for p in self.module.parameters() if p.requires_grad]

    def _groups(self):
        """(late, early) trainable parameters; without a cut everything is 'late' (one backward, one round of all-reduces)."""
        params = self._trainable()
        if not self.cut:
            return params, []
        late_ids = {id(p) for p in self.module.late_parameters()}
        return [p for p in params if id(p) in late_ids], [p for p in params if id(p) not in late_ids]

    def _build(self):
        late, early = self._groups()
        layout = tuple((id(p), p.numel()) for p in late) + ('|',) + tuple((id(p), p.numel()) for p in early)
        if layout == self.layout:
            return
        for p in self.module.parameters():
            if hasattr(p, '_sl_gview'):
                del p._sl_gview
        self.buckets, self.views = [], {}
        dev = (late + early)[0].device

        def lay_out(params):
            chunk, n = [], 0

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
        lay_out(late)
        self.late_buckets = len(self.buckets)
        lay_out(early)
        self.layout = layout
        if hasattr(self.module, 'enable_backward_cut'):
            self.module.enable_backward_cut(self.cut and bool(early))

    def adopt_gradients(self, params=None):
        """After (a half of the) backward: every .grad IS (an alias of) its bucket view.  Block backwards wrote most of them in place (functional.grad_dst);
        a gradient autograd produced itself is copied in; a parameter that received none counts as zero (DistributedDataParallel's
        find_unused_parameters=False would raise -- every parameter of the POP path receives a gradient)."""
        dst, src, zero = [], [], []
        for p in (self._trainable() if params is None else params):
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

    def all_reduce(self, which=None, async_op=False):
        """SUM over the ranks, one collective per bucket, on the process group's stream (ordered after the current stream's work).
        which: 'late' / 'early' / None (all).  async_op: returns the work handles (wait() makes the current stream wait for them)."""
        works = []
        if self.active:                                   # also at world size 1 (SEGLAND_FORCE_DDP=1): the same calls, RCCL copies
            lo, hi = (0, self.late_buckets) if which == 'late' else ((self.late_buckets, len(self.buckets)) if which == 'early' else (0, len(self.buckets)))
            for flat in self.buckets[lo:hi]:
                w = dist.all_reduce(flat, group=self.group, async_op=async_op)
                if async_op:
                    works.append(w)
        return works

    def train_step_parts(self, optimizer, double_step=True, clip_grad=5.0):
        """(backward_late(img, mask) -> loss dict, backward_early(), update_part() -> gradient norm): train_base.train_iteration cut at the all-reduces."""
        from .optim import clip_coefficient

        def backward_late(img, mask):
            self._build()
            late, early = self._groups()
            optimizer.zero_grad(set_to_none=True)
            loss_dict = self.module(img, mask)
            cuts = self.module.cut_tensors() if (self.cut and early) else None
            if cuts:
                torch.autograd.backward(loss_dict['total_loss'], inputs=[leaf for _, leaf in cuts] + late)
                self._cuts = list(cuts)
                self.adopt_gradients(late)
            else:
                loss_dict['total_loss'].backward()
                self._cuts = None
                self.adopt_gradients()
            return loss_dict

        def backward_early():
            if self._cuts:
                _, early = self._groups()
                torch.autograd.backward([t for t, _ in self._cuts], [leaf.grad for _, leaf in self._cuts], inputs=early)
                self._cuts = None
                self.module.clear_cut()               # the stash holds this step's autograd graph (and its stream-bound AccumulateGrad nodes) alive
                self.adopt_gradients(early)

        def update_part():
            params = [p for p in self._trainable() if p.grad is not None]
            norm, coef = clip_coefficient(params, clip_grad, self.world)
            optimizer.step(repeat=2 if double_step else 1, grad_scale=coef)
            return norm
        return backward_late, backward_early, update_part

    def _run(self, parts, img, mask, call, collectives=True):
        """The step with the all-reduce of the late buckets issued between the two backward halves (travelling beside the second one).
        collectives=False: the parts only, in order (GraphedBucketStep captures them back to back: nothing runs, so there is nothing to exchange)."""
        bwd1, bwd2, upd = parts
        loss = call(bwd1, img, mask)
        two = self.cut and self.late_buckets < len(self.buckets)
        works = self.all_reduce('late' if two else None, async_op=True) if collectives else []
        if two:
            call(bwd2)
            if collectives:
                works += self.all_reduce('early', async_op=True)
        for w in works:
            w.wait()
        return loss, call(upd)

    def agree(self, ok, signature=None, timeout_s=120.0):
        """True when EVERY rank says ok AND every rank is attempting the capture of the SAME input signature (one MIN all-reduce of (flag, h, -h) with h a hash of the
        signature; one host read-back, used once per capture attempt).  The ranks must reach their attempts at the same step (DistributedSampler + drop_last give every
        rank the same signature sequence); a rank whose sequence diverged would sit here while the others issue bucket all-reduces -- the wait is bounded, and what is
        raised names the cause (round-4 advisor) instead of a silent hang."""
        if not self.active:
            return bool(ok)
        import datetime
        import zlib
        h = float(zlib.crc32(repr(signature).encode()) % (1 << 22)) if signature is not None else 0.0      # exact in fp32
        flag = torch.tensor([1.0 if ok else 0.0, h, -h], dtype=torch.float32, device=self.buckets[0].device)
        try:
            work = dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.hs_group, async_op=True)
            if work.wait(datetime.timedelta(seconds=timeout_s)) is False:
                raise TimeoutError('timed out after %g s' % timeout_s)
        except (dist.DistBackendError, dist.DistNetworkError, TimeoutError) as e:
            raise RuntimeError('bucket_step: the ranks did not all reach a capture attempt at the same step (%s).  Every rank must see the same sequence of input '
                               'signatures (DistributedSampler + drop_last); start with --no-step-graph to issue the steps kernel by kernel.' % e) from e
        except RuntimeError as e:
            # the process-group backends report a timed-out or aborted collective as a plain RuntimeError; anything else (an earlier asynchronous device fault
            # surfacing at this wait) is handed on unchanged
            if not any(w in str(e).lower() for w in ('timed out', 'timeout', 'connection', 'aborted', 'socket')):
                raise
            raise RuntimeError('bucket_step: the ranks did not all reach a capture attempt at the same step (%s).  Every rank must see the same sequence of input '
                               'signatures (DistributedSampler + drop_last); start with --no-step-graph to issue the steps kernel by kernel.' % e) from e
        v = flag.tolist()                                     # (outside the handler: a device fault that surfaces at this read-back is not a rendezvous problem)
        if v[1] != -v[2]:
            import logging
            logging.getLogger('Segmentation').warning('bucket_step: the ranks attempted a capture with different input signatures; all stay kernel by kernel')
            return False
        return bool(v[0] > 0.5)

    def train_iteration(self, optimizer, img, mask, double_step=True):
        """The loop body of train_base.py:250-264 issued kernel by kernel (what GraphedBucketStep replays)."""
        return self._run(self.train_step_parts(optimizer, double_step), img, mask, lambda f, *a: f(*a))


class GraphedBucketStep:
    """Callable with the signature and results of train_base.train_iteration for a BucketedReplica: graph A1 (forward + backward down to the cut), all-reduce of the
    late buckets (asynchronous), graph A2 (rest of the backward), all-reduce of the early buckets, graph B (clip + AdamW).  The graphs are captured together (after
    `warmup` eager steps per input signature) from one memory pool WITHOUT running and without collectives; the ranks then agree on the outcome
    (BucketedReplica.agree) and either all replay -- the capturing step itself is the first replay -- or all issue this step and the following ones kernel by kernel
    (two more attempts).  Every rank must see the same sequence of input signatures (DistributedSampler + drop_last, engine._loader): the attempts line up."""

    def __init__(self, replica, optimizer, double_step=True, warmup=3):
        self.replica, self.optimizer, self.warmup = replica, optimizer, warmup
        self.parts = replica.train_step_parts(optimizer, double_step)
        self.seen, self.key, self.graphs = {}, None, None
        self.static_in, self.outs, self.static_grads = None, None, None
        self.replays, self.failures = 0, 0
        self.bn_training = True
        self.eager_reason = capturable(replica.module)
        if self.eager_reason is not None:
            import logging
            logging.getLogger('Segmentation').info('bucket_step: steps are issued kernel by kernel (%s)', self.eager_reason)

    def _state_key(self, tensors):
        m = self.replica
        return tuple((tuple(t.shape), t.dtype) for t in tensors) + (sum(1 for p in m.parameters() if p.requires_grad), sum(1 for x in m.modules() if x.training))

    def _eager(self, img, mask):
        loss, norm = self.replica._run(self.parts, img, mask, lambda f, *a: graph_step._detached(f(*a)))
        return loss, norm

    def _capture_graphs(self, img, mask):
        """-> (graphs, (loss, norm)): the parts recorded in order from one pool.  Nothing executes and no collective is issued (the unit tests replace this method)."""
        self.static_in = (img.clone(), mask.clone())
        self.optimizer.capture_begin()
        graphs, pool = [], None
        from . import functional
        functional.flush_num_batches_tracked()
        torch.cuda.synchronize()

        def call(fn, *args):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            a = self.static_in if args else ()
            with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                out = graph_step._detached(fn(*a))
            if pool is None:
                pool = g.pool()
            graphs.append(g)
            return out
        return graphs, self.replica._run(self.parts, img, mask, call, collectives=False)

    def _capture(self, img, mask, key):
        """One capture attempt on every rank.  True: self.graphs holds this step, not yet run."""
        self.graphs, err = None, None
        try:
            graphs, outs = self._capture_graphs(img, mask)
        except Exception as e:                            # noqa: BLE001  (whatever it was, the other ranks must hear about it)
            graphs, outs, err = None, None, e
            if torch.cuda.is_available():
                from . import ops
                ops.after_failed_capture()
        if not self.replica.agree(err is None, signature=key):
            self.graphs, self.key = None, None
            from . import functional
            functional.after_failed_capture()                # counters queued and caches filled by the attempt (here or on the rank that did capture): nothing of it ran
            self.failures += 1
            graph_step.STATS['failures'] += 1
            import logging
            logging.getLogger('Segmentation').warning('bucket_step: capture failed on %s (%s); every rank issues this step kernel by kernel, %s',
                                                      'this rank' if err is not None else 'another rank',
                                                      '%s: %s' % (type(err).__name__, str(err).splitlines()[0] if str(err) else '') if err is not None else 'its log has the reason',
                                                      'then another attempt' if self.failures < 3 else 'and all later ones')
            if self.failures >= 3:
                self.warmup = float('inf')
            return False
        self.graphs, self.outs, self.key = graphs, outs, key
        self.bn_training = any(isinstance(x, torch.nn.modules.batchnorm._BatchNorm) and x.training for x in self.replica.modules())
        self.static_grads = [(p, p.grad) for p in self.replica.parameters() if p.grad is not None]
        graph_step.STATS['captures'] += 1
        return True

    def __call__(self, img, mask):
        if self.eager_reason is not None:
            return self._eager(img, mask)
        key = self._state_key((img, mask))
        fresh = False
        if key != self.key or self.graphs is None:
            n = self.seen.get(key, 0)
            if n < self.warmup:
                self.seen[key] = n + 1
                return self._eager(img, mask)
            if not self._capture(img, mask, key):
                return self._eager(img, mask)             # nothing of the failed attempt has run: this is the step's one and only execution, on every rank
            fresh = True                                  # static_in holds this batch already
        if not fresh:
            for dst, src in zip(self.static_in, (img, mask)):
                dst.copy_(src, non_blocking=True)
        for p, g in self.static_grads:
            if p.grad is not g:
                p.grad = g
        it = iter(self.graphs)

        def call(fn, *args):
            if fn is self.parts[2]:
                self.optimizer.graph_prepare()
            next(it).replay()
        self.replica._run(self.parts, img, mask, call)
        self.replays += 1
        graph_step.STATS['replays'] += 1
        self._invalidate()
        return self.outs

    def _invalidate(self):
        from . import functional
        functional.weights_changed()
        if self.bn_training:
            functional.running_stats_changed()

    @property
    def graph(self):
        return self.graphs
