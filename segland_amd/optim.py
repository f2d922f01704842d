"""AdamW with the interface and state_dict layout of torch.optim.AdamW, stepped by ONE hand-written kernel launch over all parameters
(csrc/optim.hip).  `step(repeat=2)` applies the reference's two consecutive steps per iteration (train_base.py:262-264) in a single
pass over the optimizer state.  Parameters / gradients must be contiguous fp32 tensors on the GPU (there is no CPU fallback)."""
import math
import struct

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False, **unused):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError('AdamW: invalid hyper-parameter')
        if amsgrad or maximize:
            raise NotImplementedError('segland_amd.optim.AdamW: amsgrad / maximize are not implemented in the HIP kernel (use torch.optim.AdamW)')
        bad = [k for k, v in unused.items() if k not in ('foreach', 'fused', 'capturable', 'differentiable') or k in ('capturable', 'differentiable') and v]
        if bad:
            raise TypeError('segland_amd.optim.AdamW: unsupported argument(s) %s' % ', '.join(bad))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}             # launch slot -> (record bytes, device table); parameters are launched grouped by their step count
        self.repeat_next = 1          # drivers: set to 2 to fold the reference's second step() into the next one
        self._cap = None              # HIP-graph capture: launch slot -> persistent {pinned table, device table, device hyper-parameters}
        self._graph_plan = []         # [(slot, parameters, repeat)] of the captured step(), replayed by graph_prepare()

    @torch.no_grad()
    def step(self, closure=None, repeat=None, grad_scale=None):
        """grad_scale: optional 1-element float32 GPU tensor multiplied into every gradient inside the kernel (the clip_grad_norm_
        coefficient, see clip_coefficient()) -- the gradients themselves are left unscaled.
        torch.optim.AdamW keeps a step count per parameter (a parameter that first receives a gradient later, or a resumed state_dict with
        mixed counts): parameters are grouped by count, one launch per distinct count -- one launch in the usual case."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        repeat = self.repeat_next if repeat is None else repeat
        self.repeat_next = 1
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing and self._cap is None:
            raise RuntimeError('segland_amd.optim.AdamW: call capture_begin() before capturing step() into a graph (segland_amd.graph_step does)')
        groups, hyper, dev = {}, None, None          # step count -> [record bytearray, n, chunk start, parameters]
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group['betas']
            hp = (b1, b2, group['eps'])
            if hyper is not None and hp != hyper:
                raise RuntimeError('segland_amd.optim.AdamW: betas / eps must be equal across parameter groups')
            hyper = hp
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise RuntimeError('segland_amd.optim.AdamW: contiguous float32 GPU parameters and gradients only')
                dev = p.device
                st = self.state[p]
                if not st:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t = int(st['step']) + 1
                if not capturing:                   # a captured step does not execute: graph_prepare() counts the replays
                    st['step'] += repeat
                g = groups.setdefault(t, [bytearray(), 0, 0, []])
                g[0] += struct.pack('<QQQQqffqii', p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                                    p.numel(), group['lr'], group['weight_decay'], g[2], gi, 0)
                g[2] += (p.numel() + 4095) // 4096
                g[1] += 1
                g[3].append(p)
        if not groups:
            return loss
        b1, b2, eps = hyper
        for slot, (step_t, (rec, n, total, plist)) in enumerate(sorted(groups.items())):
            if capturing:
                cap = self._cap.get(slot)
                if cap is None or len(rec) > cap['pinned'].numel():
                    raise RuntimeError('segland_amd.optim.AdamW: the captured step touches more parameters / step counts than capture_begin() saw')
                cap['pinned'][:len(rec)] = torch.frombuffer(rec, dtype=torch.uint8)        # host write now; the H2D copy below is a graph node
                cap['dev'].copy_(cap['pinned'], non_blocking=True)
                ops.adamw_multi_dev(cap['dev'], n, total, b1, b2, eps, cap['hyp'], repeat, grad_scale=grad_scale)
                self._graph_plan.append((slot, plist, repeat))
                continue
            ent = self._tables.get(slot)
            if ent is None or ent[0] != rec:      # the caching allocator hands back the same gradient addresses step after step: usually no upload
                # pinned staging + asynchronous copy: a pageable upload would synchronise the host with the GPU once per step
                ent = self._tables[slot] = (bytes(rec), torch.frombuffer(bytearray(rec), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True))
            bc = [(1.0 - b1 ** (step_t + r), math.sqrt(1.0 - b2 ** (step_t + r))) for r in (0, 1)]
            ops.adamw_multi(ent[1], n, total, b1, b2, eps, bc[0][0], bc[0][1], bc[1][0], bc[1][1], repeat, grad_scale=grad_scale)
        return loss


    # ---- whole-step HIP graphs (segland_amd/graph_step.py) ---------------------------------------------------------------------------
    def capture_begin(self):
        """Before capturing a step(): allocates what a captured launch needs to stay valid across replays -- per launch slot a pinned host
        table + its device copy (gradient addresses are only known inside the capture) and the device vector of step-dependent scalars."""
        dev, counts, n = None, set(), 0
        for group in self.param_groups:
            for p in group['params']:
                if p.requires_grad and p.is_cuda:
                    dev = p.device
                    n += 1
                    st = self.state[p]
                    if not st:
                        # materialise the moments NOW: allocated inside the capture they would live in the graph's pool and their zero-fill
                        # would be a graph node -- every replay would reset them (GraphedStep(warmup=0), or a parameter whose first gradient
                        # arrives at capture time)
                        st['step'] = torch.tensor(0.0)
                        st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    counts.add(int(st['step']))
        if dev is None:
            raise RuntimeError('segland_amd.optim.AdamW.capture_begin: no GPU parameters')
        self._cap, self._graph_plan = {}, []
        for slot in range(max(1, len(counts))):
            self._cap[slot] = dict(pinned=torch.zeros(64 * n, dtype=torch.uint8).pin_memory(), dev=torch.zeros(64 * n, dtype=torch.uint8, device=dev),
                                   hyp=torch.zeros(4 + 2 * len(self.param_groups), dtype=torch.float32, device=dev))

    def graph_prepare(self):
        """Before every replay of a graph that holds a captured step(): advances the step counts like step() would and puts this step's bias corrections and the
        groups' current lr / weight_decay into the device vector the captured kernel reads.
        NOT by a host-to-device copy per step: measured (profiles/r2_kernel_sequence_graph_step.txt, tools/graph_host_time.py), the 40-byte upload sat on the
        critical path of every step with 0.3-0.5 ms of GPU idle time in front of it -- the copy engine's hand-off behind the previous graph -- although the host
        had issued it 25 ms earlier.  The rows of the next 1024 steps (host arithmetic, same values as step() computes) live in a DEVICE table that is uploaded
        once and refreshed when it runs out or a group's lr / weight_decay changes; per step one row moves device-to-device (a 5 us kernel on the compute queue)."""
        for slot, plist, repeat in self._graph_plan:
            t = int(self.state[plist[0]]['step']) + 1
            for p in plist:
                self.state[p]['step'] += repeat
            b1, b2 = self.param_groups[0]['betas']
            cap = self._cap[slot]
            sig = tuple((float(g['lr']), float(g['weight_decay'])) for g in self.param_groups) + (b1, b2, repeat)
            tab = cap.get('tab')
            if tab is None or tab['sig'] != sig or t < tab['t0'] or t >= tab['t0'] + tab['n'] * repeat or (t - tab['t0']) % repeat:
                n = 1024
                tail = [v for g in self.param_groups for v in (g['lr'], g['weight_decay'])]
                rows = [[1.0 - b1 ** tt, math.sqrt(1.0 - b2 ** tt), 1.0 - b1 ** (tt + 1), math.sqrt(1.0 - b2 ** (tt + 1))] + tail for tt in range(t, t + n * repeat, repeat)]
                tab = cap['tab'] = dict(sig=sig, t0=t, n=n, dev=torch.tensor(rows, dtype=torch.float32).pin_memory().to(cap['hyp'].device, non_blocking=True))
            cap['hyp'].copy_(tab['dev'][(t - tab['t0']) // repeat], non_blocking=True)


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD (momentum, weight decay; the reference's fine-tuning optimizer, ft_pop.py:205-209) with torch's interface and state_dict layout
    (`momentum_buffer`), stepped by ONE kernel launch over all parameters (csrc/optim.hip sl_sgd_multi) and capturable: inside a HIP graph the kernel reads each
    group's (lr, weight_decay) from device memory, which graph_prepare() refreshes before every replay -- ft_pop changes the learning rate every iteration
    (ft_pop.py:246-249), which is what kept torch's SGD (it bakes lr into its launches: four small launches per step) outside the captured step until round 4.
    dampening / nesterov / maximize are not implemented (the reference uses none of them)."""
    supports_grad_scale = True      # step(grad_scale=coef): the clip_grad_norm_ coefficient applied inside the kernel (utils.pyt_utils.NativeScalerWithGradNormCount, ft_pop.ft_graph_body)

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, maximize=False, **unused):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError('SGD: invalid hyper-parameter')
        if dampening or nesterov or maximize:
            raise NotImplementedError('segland_amd.optim.SGD: dampening / nesterov / maximize are not implemented in the HIP kernel (use torch.optim.SGD)')
        bad = [k for k, v in unused.items() if k not in ('foreach', 'fused', 'differentiable') or k == 'differentiable' and v]
        if bad:
            raise TypeError('segland_amd.optim.SGD: unsupported argument(s) %s' % ', '.join(bad))
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=False))
        self._table = None            # (record bytes, device table) of the last eager launch
        self._cap = None              # HIP-graph capture: persistent {pinned table, device table, device (lr, wd) vector}
        self._captured = False

    def _records(self, capturing=False):
        """The per-parameter launch records.  (lr, weight_decay) are NOT in them: the kernel reads both from the device vector `hyper` (one entry pair per group), so the
        records -- and the device table built from them -- change only when a pointer does (round-4 advisor: ft_pop changes lr every iteration, which used to rebuild the
        table, pin a fresh host tensor and copy it on every kernel-by-kernel step)."""
        rec, n, chunks, mom, dev = bytearray(), 0, 0, None, None
        for gi, group in enumerate(self.param_groups):
            if mom is not None and group['momentum'] != mom:
                raise RuntimeError('segland_amd.optim.SGD: momentum must be equal across parameter groups')
            mom = group['momentum']
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise RuntimeError('segland_amd.optim.SGD: contiguous float32 GPU parameters and gradients only')
                dev = p.device
                buf = 0
                if mom:
                    st = self.state[p]
                    if 'momentum_buffer' not in st or st['momentum_buffer'] is None:
                        if capturing:        # an allocation + fill recorded into the graph would zero the buffer again on every replay
                            raise RuntimeError('segland_amd.optim.SGD: a parameter received its first gradient inside a captured step (requires_grad toggled after '
                                               'capture_begin()?); capture again')
                        st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)      # torch: buf = clone(d) on the first step == 0 * momentum + d
                    buf = st['momentum_buffer'].data_ptr()
                rec += struct.pack('<QQQQqffqii', p.data_ptr(), p.grad.data_ptr(), buf, 0, p.numel(), 0.0, 0.0, chunks, gi, 0)
                chunks += (p.numel() + 4095) // 4096
                n += 1
        return rec, n, chunks, mom or 0.0, dev

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if capturing and self._cap is None:
            raise RuntimeError('segland_amd.optim.SGD: call capture_begin() before capturing step() into a graph (segland_amd.graph_step does)')
        rec, n, chunks, mom, dev = self._records(capturing)
        if n == 0:
            return loss
        if capturing:
            cap = self._cap
            if len(rec) > cap['pinned'].numel():
                raise RuntimeError('segland_amd.optim.SGD: the captured step touches more parameters than capture_begin() saw')
            cap['pinned'][:len(rec)] = torch.frombuffer(rec, dtype=torch.uint8)          # host write now; the H2D copy below is a graph node
            cap['dev'].copy_(cap['pinned'], non_blocking=True)
            ops.sgd_multi(cap['dev'], n, chunks, mom, hyper=cap['hyp'], grad_scale=grad_scale)
            self._captured = True
            return loss
        ent = self._table
        if ent is None or ent[0] != rec:
            if 2 * len(self.param_groups) > 16:
                raise RuntimeError('segland_amd.optim.SGD: at most 8 parameter groups')
            ent = self._table = (bytes(rec), torch.frombuffer(bytearray(rec), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True),
                                 torch.zeros(16, dtype=torch.float32, device=dev))
        ops.store_floats(ent[2], [v for g in self.param_groups for v in (g['lr'], g['weight_decay'])])      # kernel arguments of a one-block launch: no host -> device copy
        ops.sgd_multi(ent[1], n, chunks, mom, hyper=ent[2], grad_scale=grad_scale)
        return loss

    # ---- whole-step HIP graphs (segland_amd/graph_step.py): same protocol as AdamW
    def capture_begin(self):
        dev, n = None, 0
        for group in self.param_groups:
            for p in group['params']:
                if p.requires_grad and p.is_cuda:
                    dev, n = p.device, n + 1
                    if group['momentum'] and self.state[p].get('momentum_buffer') is None:
                        self.state[p]['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)      # not inside the capture: a replay would zero it again
        if dev is None:
            raise RuntimeError('segland_amd.optim.SGD.capture_begin: no GPU parameters')
        if 2 * len(self.param_groups) > 16:
            raise RuntimeError('segland_amd.optim.SGD: at most 8 parameter groups in a captured step')
        self._cap = dict(pinned=torch.zeros(64 * n, dtype=torch.uint8).pin_memory(), dev=torch.zeros(64 * n, dtype=torch.uint8, device=dev),
                         hyp=torch.zeros(16, dtype=torch.float32, device=dev))
        self._captured = False

    def graph_prepare(self):
        """Before every replay of a graph that holds a captured step(): this iteration's (lr, weight_decay) per group -> the device vector the kernel reads, as
        kernel arguments of a one-block launch (no host -> device copy in front of the graph)."""
        if self._cap is not None and self._captured:
            ops.store_floats(self._cap['hyp'], [v for g in self.param_groups for v in (g['lr'], g['weight_decay'])])


_NORM_KERNEL = True      # test hook: clip_coefficient through csrc/optim.hip's gradient-norm kernels (False: torch's _foreach_norm chain)


def clip_coefficient(parameters, max_norm, grad_div=1):
    """(total_norm, coefficient) of torch.nn.utils.clip_grad_norm_(parameters, max_norm) WITHOUT scaling the gradients: the coefficient
    min(1, max_norm / (total_norm + 1e-6)) is handed to AdamW.step(grad_scale=...) and applied inside the optimizer kernel.
    grad_div > 1: the gradients hold the SUM over that many ranks (engine.enable_inplace_bucket_gradients); the norm is that of their mean
    and the returned coefficient carries the 1 / grad_div as well."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0), None
    if _NORM_KERNEL and all(g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() for g in grads):
        # the whole of it -- sum of squares over every gradient, square root, coefficient -- in ceil(n / 64) + 1 launches (round 6; torch: ~12 launches, 130 us per ResNet-50 step)
        return ops.grad_norm_coef(grads, max_norm, grad_div)
    total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
    if grad_div != 1:
        total = total / grad_div
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0).to(torch.float32).reshape(1)
    if grad_div != 1:
        coef = coef / grad_div
    return total, coef
