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


def eligible(model, optimizer, device):
    from .optim import AdamW
    return (os.environ.get('SEGLAND_STEP_GRAPH', '1') != '0' and torch.cuda.is_available() and torch.device(device).type == 'cuda'
            and isinstance(optimizer, AdamW) and not isinstance(model, torch.nn.parallel.DistributedDataParallel))


class GraphedTrainStep:
    """Callable with the signature and results of train_base.train_iteration(model, optimizer, loss_scaler, img, mask).
    The first `warmup` calls per input shape run eagerly (lazy allocations, optimizer state); the next one is captured; later calls copy
    the batch into the graph's static inputs and replay.  The returned loss dict / gradient norm are the graph's static outputs: read
    them (`.item()`, `float()`) before the next call."""

    def __init__(self, step_fn, model, optimizer, loss_scaler, double_step=True, warmup=3):
        self.step_fn, self.model, self.optimizer, self.loss_scaler = step_fn, model, optimizer, loss_scaler
        self.double_step, self.warmup = double_step, warmup
        self.seen, self.key, self.graph = {}, None, None
        self.static_in, self.static_out, self.static_grads = None, None, None
        self.replays, self.failures = 0, 0

    def _state_key(self, img, mask):
        trainable = sum(1 for p in self.model.parameters() if p.requires_grad)
        modes = sum(1 for m in self.model.modules() if m.training)
        return (tuple(img.shape), img.dtype, tuple(mask.shape), mask.dtype, trainable, modes)

    def _eager(self, img, mask):
        out = self.step_fn(self.model, self.optimizer, self.loss_scaler, img, mask, double_step=self.double_step)
        # detached: a loss the caller keeps would keep this step's autograd graph -- and its AccumulateGrad nodes, bound to THIS stream --
        # alive into the capture on another stream (autograd then synchronises the two streams, which a capture cannot contain)
        d = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out[0].items()} if isinstance(out[0], dict) else out[0]
        return (d,) + tuple(out[1:])

    def _capture(self, img, mask, key):
        self.graph = None                                   # drop an older graph (and its pool) first
        self.static_in = (img.clone(), mask.clone())
        self.optimizer.capture_begin()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            out = self._eager(*self.static_in)
        self.graph, self.key, self.static_out = g, key, out
        # the gradients the replays write: tensors of the graph's pool that the parameters keep pointing at
        self.static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]

    def __call__(self, img, mask):
        key = self._state_key(img, mask)
        if key != self.key:
            n = self.seen.get(key, 0)
            if n < self.warmup:
                self.seen[key] = n + 1
                return self._eager(img, mask)
            try:
                self._capture(img, mask, key)
            except Exception as e:              # something on the path still needed the host (a lazy upload, a read-back): run it eagerly
                self.graph, self.key = None, None
                self.failures += 1
                torch.cuda.synchronize()
                import logging
                logging.getLogger('Segmentation').warning('graph_step: capture failed (%s: %s); %s', type(e).__name__, str(e).splitlines()[0] if str(e) else '',
                                                          'one more eager step, then another attempt' if self.failures < 3 else 'staying eager')
                if self.failures >= 3:
                    self.warmup = float('inf')
                return self._eager(img, mask)
        self.static_in[0].copy_(img, non_blocking=True)
        self.static_in[1].copy_(mask, non_blocking=True)
        for p, g in self.static_grads:                      # an eager step in between (odd last batch) re-pointed .grad
            if p.grad is not g:
                p.grad = g
        self.optimizer.graph_prepare()
        self.graph.replay()
        self.replays += 1
        return self.static_out
