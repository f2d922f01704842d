"""The training step as ONE HIP graph: forward, loss, backward, gradient-norm clip and both AdamW steps of the loop body
(train_base.py:250-264) are captured once per input shape and replayed, so a step costs one graph launch instead of ~750 (PSPNet-POP)
/ ~1300 (Swin-POP) kernel launches issued from Python.  At 16 tiles per GPU the ResNet step is GPU-bound either way; at the 8 tiles per
GPU of the Swin configuration (BASELINE config 5) and in fine-tuning the launches were the bottleneck.

What makes the step replayable: every kernel of the path takes its stream from torch (ops._s), nothing on the path synchronises or reads
a value back, BatchNorm's `num_batches_tracked` and running statistics are updated by kernels, DropPath / Dropout2d draw from torch's
graph-safe Philox state, and segland_amd.optim.AdamW keeps its step-dependent scalars (bias corrections, the groups' lr and weight
decay) in device memory that `graph_prepare()` refreshes before each replay.  Not captured: DistributedDataParallel (its reducer and
RCCL work run eagerly) -- `eligible()` says no and the caller keeps `train_iteration`."""
import os

import torch


STATS = {'captures': 0, 'replays': 0, 'failures': 0}          # process-wide counters (tests / logs: did the drivers really replay?)


def eligible(model, optimizer, device, need_adamw=True):
    from .optim import AdamW
    return (os.environ.get('SEGLAND_STEP_GRAPH', '1') != '0' and torch.cuda.is_available() and torch.device(device).type == 'cuda'
            and (isinstance(optimizer, AdamW) or not need_adamw) and not isinstance(model, torch.nn.parallel.DistributedDataParallel))


def _detached(out):
    """Losses handed to the caller without their autograd graph: a loss the caller keeps would keep this step's graph -- and its
    AccumulateGrad nodes, bound to THIS stream -- alive into a capture on another stream (autograd then synchronises the two streams,
    which a capture cannot contain)."""
    if isinstance(out, dict):
        return {k: _detached(v) for k, v in out.items()}
    if isinstance(out, (tuple, list)):
        return type(out)(_detached(v) for v in out)
    return out.detach() if torch.is_tensor(out) else out


class GraphedStep:
    """`body(*tensors)` -- a function of GPU tensors that launches the kernels of a step and returns tensors -- captured once per input
    signature and replayed.  The first `warmup` calls per signature run eagerly (lazy allocations, optimizer state, weight-preparation
    tables); the next one is captured; later calls copy the arguments into the graph's static inputs and replay.  The returned tensors
    are the graph's static outputs: read them (`.item()`, `float()`) before the next call.  `optimizer`: a segland_amd.optim.AdamW whose
    step() is inside `body` (its step counts and hyper-parameters are advanced / uploaded before every replay) or None."""

    def __init__(self, body, model, optimizer=None, warmup=3):
        self.body, self.model, self.optimizer, self.warmup = body, model, optimizer, warmup
        self.seen, self.key, self.graph = {}, None, None
        self.static_in, self.static_out, self.static_grads = None, None, None
        self.replays, self.failures, self.baked = 0, 0, None
        self.bn_training = True

    def _state_key(self, tensors):
        trainable = sum(1 for p in self.model.parameters() if p.requires_grad)
        modes = sum(1 for m in self.model.modules() if m.training)
        from . import functional
        # ft mode: the frozen base classifier's prototype rows are cached tensors baked into a captured step (functional._base_chain): what they are computed from is
        # part of the key, so a weight change behind the graph (load_state_dict, init_cls_n) re-captures instead of replaying stale rows
        return tuple((tuple(t.shape), t.dtype) for t in tensors) + (trainable, modes, functional.base_chain_key(self.model))

    def _eager(self, tensors):
        return _detached(self.body(*tensors))

    def _capture(self, tensors, key):
        self.graph = None                                   # drop an older graph (and its pool) first
        self.static_in = tuple(t.clone() for t in tensors)
        if self.optimizer is not None:
            self.optimizer.capture_begin()
        from . import functional
        functional.flush_num_batches_tracked()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            out = self._eager(self.static_in)
        self.graph, self.key, self.static_out = g, key, out
        self.baked = self.model.__dict__.get('_sl_base_chain')      # tensors of the cache entry the captured launches read: alive as long as this graph is
        self.bn_training = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training for m in self.model.modules())
        STATS['captures'] += 1
        # the gradients the replays write: tensors of the graph's pool that the parameters keep pointing at
        self.static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]

    def __call__(self, *tensors):
        key = self._state_key(tensors)
        if key != self.key:
            n = self.seen.get(key, 0)
            if n < self.warmup:
                self.seen[key] = n + 1
                return self._eager(tensors)
            try:
                self._capture(tensors, key)
            except Exception as e:              # something on the path still needed the host (a lazy upload, a read-back): run it eagerly
                self.graph, self.key = None, None
                self.failures += 1
                STATS['failures'] += 1
                from . import functional, ops
                ops.after_failed_capture()
                functional.after_failed_capture()
                import logging
                logging.getLogger('Segmentation').warning('graph_step: capture failed (%s: %s); %s', type(e).__name__, str(e).splitlines()[0] if str(e) else '',
                                                          'one more eager step, then another attempt' if self.failures < 3 else 'staying eager')
                if self.failures >= 3:
                    self.warmup = float('inf')
                return self._eager(tensors)
        for dst, src in zip(self.static_in, tensors):
            dst.copy_(src, non_blocking=True)
        for p, g in self.static_grads:                      # an eager step in between (odd last batch) or zero_grad() re-pointed .grad
            if p.grad is not g:
                p.grad = g
        if self.optimizer is not None:
            self.optimizer.graph_prepare()
        self.graph.replay()
        self.replays += 1
        STATS['replays'] += 1
        # A replay runs no Python: the optimizer kernel, the BN running-statistics kernels and the weight re-preparation (which sits BEFORE the
        # optimizer in the captured order, so the prepared copies lag the parameters by one step) all went through raw pointers.  Invalidate every
        # host-side cache keyed on them (prepared weights, BN eval coefficients, the captured eval feature graph): the next eager forward --
        # validation between epochs -- must see this replay's weights and statistics.
        from . import functional
        if self.optimizer is not None:
            functional.weights_changed()
        if self.bn_training:
            functional.running_stats_changed()
        return self.static_out


class GraphedTrainStep(GraphedStep):
    """Callable with the signature and results of train_base.train_iteration(model, optimizer, loss_scaler, img, mask): the whole loop body
    (train_base.py:250-264) including clip + both AdamW steps in the graph."""

    def __init__(self, step_fn, model, optimizer, loss_scaler, double_step=True, warmup=3):
        super().__init__(lambda img, mask: step_fn(model, optimizer, loss_scaler, img, mask, double_step=double_step), model, optimizer, warmup)
