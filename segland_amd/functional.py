"""Block-level autograd Functions: each one runs a fused chain of libsegland_hip.so kernels forward and the
hand-written backward chain, so PyTorch's autograd only links blocks together and accumulates parameter gradients
(which is what lets DDP's bucketed RCCL all-reduce overlap with the backward of earlier blocks).

nn.Conv2d / nn.BatchNorm2d modules are used purely as parameter holders (state_dict compatibility with the
reference); their own forward is never called.
"""
import os

import torch
from torch.autograd.function import once_differentiable

from . import ops
from .ops import ConvSpec

_nbt_pending = []       # num_batches_tracked buffers to bump once per forward

# Staleness of the derived (GEMM-layout / compute-dtype) weight copies.  The tensor version counter alone is not enough: fused
# optimizers (torch.optim.AdamW(fused=True), the drivers' default on the GPU) update parameters WITHOUT bumping `_version`.  A global
# post-step hook on every torch optimizer advances an epoch that is part of the cache key of trainable weights; frozen weights
# (requires_grad False: the ft_pop backbone) keep their copies across steps, load_state_dict still bumps their version.
_OPT_EPOCH = [0]


def _on_optimizer_step(optimizer, args, kwargs):
    _OPT_EPOCH[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_step      # noqa: E402

_register_post_step(_on_optimizer_step)


def weights_changed():
    """For code that mutates parameters behind torch's back (`.data` writes, custom optimizers that are not torch.optim subclasses,
    HIP-graph replays that contain an optimizer step: no Python hook runs during a replay)."""
    _OPT_EPOCH[0] += 1


# BatchNorm running statistics are written by kernels through raw pointers.  An eager train-mode forward bumps the module's own
# `_sl_rs_epoch`; a HIP-graph replay runs no Python at all, so graph_step bumps this global epoch after every replay.  It is part of
# the key of every cache derived from running statistics (_bn_eval_coeffs, GFSS_Model._features_graphed).
_RS_EPOCH = [0]


def running_stats_changed():
    _RS_EPOCH[0] += 1


def _wver(w):
    if w is None:
        return None
    return (w._version, _OPT_EPOCH[0] if w.requires_grad else 0)


def prepared(w, dtype):
    """GEMM-layout copies of a conv weight in the compute dtype, refreshed when the parameter changes.
    The cache lives on the Parameter object itself (version counter + storage pointer + dtype as the key)."""
    ent = getattr(w, '_sl_prep', None)
    if ent is None or ent[0] != _wver(w) or ent[1] != dtype or ent[2] != w.data_ptr():
        wf, wb = ops.weight_prep(w, dtype)
        ent = (_wver(w), dtype, w.data_ptr(), wf, wb)
        w._sl_prep = ent
    return ent[3], ent[4]


class _PrepPlan:
    """All conv weights of a model -> GEMM layouts in one kernel launch per optimizer step (instead of one per conv)."""

    def __init__(self):
        self.key, self.table, self.total, self.items = None, None, 0, []

    def refresh(self, convs, dtypes):
        import struct
        ws = [c.weight for c in convs]
        stale = [w for w, d in zip(ws, dtypes) if (getattr(w, '_sl_prep', None) is None or w._sl_prep[0] != _wver(w)
                                                    or w._sl_prep[1] != d or w._sl_prep[2] != w.data_ptr())]
        if not stale:
            return
        if len(stale) * 4 < len(ws) and self.key is not None:   # a few trainable weights over a frozen backbone (ft_pop): per-conv launches
            for w, d in zip(ws, dtypes):
                if w._sl_prep[0] != _wver(w) or w._sl_prep[1] != d or w._sl_prep[2] != w.data_ptr():
                    prepared(w, d)
            return
        key = tuple((w.data_ptr(), d) for w, d in zip(ws, dtypes))
        if key != self.key:                                   # (re)build buffers + the device table
            rec, start, self.items = b'', 0, []
            for w, d in zip(ws, dtypes):
                O, I, KH, KW = w.shape
                wf = torch.empty((O, KH, KW, I), dtype=d, device=w.device)
                wb = torch.empty((I, KH, KW, O), dtype=d, device=w.device)
                rec += struct.pack('<QQQiiiiq', w.data_ptr(), wf.data_ptr(), wb.data_ptr(), O, I, KH * KW, ops.dt(d), start)
                assert O % 64 == 0 and I % 32 == 0 and KH * KW <= 9
                start += O * I // 2048
                self.items.append((w, d, wf, wb))
            self.table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(ws[0].device)
            self.total, self.key = start, key
        ops.weight_prep_batched(self.table, len(self.items), self.total,
                                nbytes=sum(w.numel() * (4 + 2 * wf.element_size()) for w, _, wf, _ in self.items))
        for w, d, wf, wb in self.items:
            w._sl_prep = (_wver(w), d, w.data_ptr(), wf, wb)


def refresh_weights(model):
    """Called once per forward by GFSS_Model: re-derives the GEMM-layout copies of every conv weight whose version changed."""
    plan = model.__dict__.get('_sl_prep_plan')
    if plan is None:
        plan = model.__dict__['_sl_prep_plan'] = _PrepPlan()
        convs, dtypes = [], []
        stage_convs = {id(st[1]) for st in model.decoder.stages}
        for m in list(model.backbone.modules()) + list(model.decoder.modules()) + list(model.classifier.modules()) + \
                (list(model.classifier_n.modules()) if getattr(model, 'classifier_n', None) is not None else []):
            if isinstance(m, torch.nn.Conv2d) and m.kernel_size[0] in (1, 3) and m.out_channels % 64 == 0 and m.in_channels % 32 == 0 \
                    and not (_PPM_FACTORISED and m is model.decoder.bottleneck[0]):       # that one is consumed as slices (_ppm_weights)
                convs.append(m)
                dtypes.append(torch.float32 if id(m) in stage_convs else model.compute_dtype)     # PPM stage path is fp32
        plan.convs, plan.dtypes = convs, dtypes
    if all(w.weight.is_cuda and w.weight.dtype == torch.float32 and w.weight.is_contiguous() for w in plan.convs):
        plan.refresh(plan.convs, plan.dtypes)


# ---- gradients written in place into DDP's bucket views --------------------------------------------------------------------------------
# Under DistributedDataParallel(gradient_as_bucket_view=True) a parameter's .grad is a view into the flat bucket the RCCL all-reduce runs on.
# The engine caches those views on the parameters (`_sl_gview`, refreshed before every optimizer step); a block backward that finds one
# lets its kernels write the gradient THERE and hands autograd a fresh alias of it: AccumulateGrad adopts the alias without a copy and
# DDP's reducer, seeing `grad.is_alias_of(bucket_view)`, neither copies nor (with the engine's sum-only comm hook) scales it -- the ~290
# per-parameter copy / scale kernels of a plain DDP step disappear.  Correct whatever the cache holds: DDP compares storages itself and
# falls back to its copy when the view is stale (the one iteration after it rebuilt its buckets).
def grad_dst(p):
    v = getattr(p, '_sl_gview', None)
    if v is None or v.shape != p.shape or v.dtype != torch.float32 or not v.is_contiguous() or p.grad is not None:
        return None                                    # p.grad set already (gradient accumulation): autograd must ADD, so no in-place write
    return v


def grad_alias(t, dst):
    """What a backward returns for a gradient it wrote into `dst` (a cached bucket view): a new tensor object on the same storage."""
    return t.detach() if (dst is not None and t is dst) else t


def flush_num_batches_tracked():
    """One launch for the counters of every train-mode BatchNorm that ran since the last flush (the models call it at the end of their feature extractor).
    The step drivers also call it right BEFORE they start a graph capture: an entry left behind by code that ran BatchNorm layers outside a model's forward (unit
    tests of single blocks) would otherwise be bumped by a node of the new graph -- on every replay, for as long as the graph lives, whatever has become of the
    module that owned the counter."""
    if _nbt_pending:
        torch._foreach_add_(_nbt_pending, 1)
        _nbt_pending.clear()


def discard_pending_counters():
    """After a FAILED graph capture: the BatchNorm layers of the attempt queued their counters, but nothing of the attempt ran (the step is about to be issued again,
    kernel by kernel, and queues them again)."""
    _nbt_pending.clear()


def after_failed_capture():
    """Host-side state a failed capture attempt leaves behind: queued BatchNorm counters (nothing of the attempt ran) and cache entries whose CONTENT was to be produced
    by captured launches -- prepared weight copies, BN evaluation coefficients -- under keys that would still match in the eager step that follows."""
    discard_pending_counters()
    weights_changed()
    running_stats_changed()


def spec_of(conv):
    return ConvSpec(conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0])


def _bn_coeffs(bn, part, count):
    """(mean, invstd, scale, shift) for this call; train mode also updates the running statistics."""
    if bn.training:
        world = sync_world(bn)
        # torch checks the GLOBAL count under SyncBatchNorm (nn/modules/_functions.py) and the local one under BatchNorm2d
        # (F.batch_norm): with SEGLAND_SYNC_BN=1 a per-GPU batch of 1 (PPM level 1: B*1*1 values per channel) trains, without it raises
        if count * max(world, 1) <= 1:
            raise ValueError('Expected more than 1 value per channel when training, got %d (per-GPU batch 1 needs SEGLAND_SYNC_BN=1: '
                             'the default BatchNorm statistics are per GPU)' % count)
        if bn.momentum is None:
            raise RuntimeError('segland_amd: cumulative-average BatchNorm (momentum=None) is not supported')
        if world:                                            # SyncBatchNorm: global sum / sum of squares / count (equal shards)
            part, count = ops.allreduce_partials(part), count * world
        out = ops.bn_finalize_train(part, count, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        _nbt_pending.append(bn.num_batches_tracked)
        # the kernel wrote the running statistics through raw pointers (no tensor version bump): invalidate the eval-coefficient cache
        bn.__dict__['_sl_rs_epoch'] = bn.__dict__.get('_sl_rs_epoch', 0) + 1
        return out
    return _bn_eval_coeffs(bn)


_SYNC_BN = os.environ.get('SEGLAND_SYNC_BN', '0')


def set_sync_bn(mode):
    """'0' (per-GPU statistics, default), '1' (synchronise nn.SyncBatchNorm modules when world_size > 1) or 'force' (also at world_size 1)."""
    global _SYNC_BN
    _SYNC_BN = str(mode)


def sync_world(bn):
    """World size if this BN synchronises its batch statistics over the process group, else 0.  Default: statistics are per GPU
    (DESIGN.md section 6); SEGLAND_SYNC_BN=1 gives nn.SyncBatchNorm modules the reference's distributed semantics (train_base.py:175-176)
    at the price of two small all-reduces per layer and direction."""
    if _SYNC_BN == '0' or not bn.training or not isinstance(bn, torch.nn.SyncBatchNorm):
        return 0
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    w = dist.get_world_size()
    return w if (w > 1 or _SYNC_BN == 'force') else 0


def conv_bn_fwd(x, conv, bn, relu, residual=None, x2=None, out=None, want_mask=False):
    wf, _ = prepared(conv.weight, x.dtype)
    c, part = ops.conv2d_fwd(x, wf, spec_of(conv), x2=x2, want_stats=bn.training)
    mean, invstd, scale, shift = _bn_coeffs(bn, part, c.numel() // c.shape[-1])
    if want_mask:
        y, mask = ops.bn_act(c, scale, shift, residual=residual, relu=relu, out=out, want_mask=True)
        return c, y, mean, invstd, mask
    y = ops.bn_act(c, scale, shift, residual=residual, relu=relu, out=out)
    return c, y, mean, invstd


def conv_bn_infer(x, conv, bn, relu, residual=None, x2=None, out=None):
    """Frozen-statistics conv+BN(+residual)(+ReLU) as ONE kernel (no conv-output round trip, nothing saved)."""
    wf, _ = prepared(conv.weight, x.dtype)
    _, _, scale, shift = _bn_eval_coeffs(bn)
    return ops.conv2d_affine_fwd(x, wf, spec_of(conv), scale, shift, x2=x2, residual=residual, relu=relu, out=out)


def _bn_eval_coeffs(bn):
    """(mean, invstd, scale, shift) of a BN on its running statistics, cached on the module until any of its four tensors changes
    (58 tiny launches per frozen forward otherwise -- the ft_pop step is launch-bound)."""
    key = (_wver(bn.weight), _wver(bn.bias), bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(), bn.eps,
           bn.__dict__.get('_sl_rs_epoch', 0), _RS_EPOCH[0])      # train-mode forwards / graph replays update the statistics behind the version counters
    ent = bn.__dict__.get('_sl_eval')
    if ent is None or ent[0] != key:
        ent = (key, ops.bn_finalize_eval(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
        bn.__dict__['_sl_eval'] = ent
    return ent[1]


def _frozen(ctx, *bns):
    """True when nothing in this block needs a gradient and every BN runs on its running statistics."""
    return not any(ctx.needs_input_grad) and not any(b.training for b in bns)


_PPM_WGRAD_GROUPED = True  # test hook: the pyramid's eight row-GEMM weight gradients as two grouped launches (ops.ppm_rows_wgrad); False: one generic weight-gradient launch + slab reduce per level
_STAGE_BN_GROUPED = True   # test hook: the pyramid stages' BatchNorm backward in one launch (ops.ppm_stage_bn_bwd); SyncBatchNorm stages always take the per-level chain
_BN3_FOLD = False          # test hook (default OFF: built and measured 678.3 vs 679.7 tiles/s, profiles/r6_ab_bn3_fold.txt): bn3's backward apply pass folded into conv3's data and weight gradient where the incoming gradient arrived gated and reduced
_WGRAD_BATCH = False       # test hook: the flat slab reduces of a bottleneck's 1x1 weight gradients in one launch (ops.WgradBatch).  Measured NEGATIVE on the ResNet-50 step (691.1 vs 693.2 tiles/s, profiles/r6_ab_r50_wbatch.txt: the deferred reduce reads cold slabs); kept for the Swin blocks, whose slabs are small
_DS_HALF = True            # test hook: the data gradient of a stride-2 1x1 downsample conv stays on its own grid (conv2d_bwd_data_addend_half)
_BASE_CHAIN_CACHE = True   # test hook: ft mode, the frozen base classifier's rows are computed once (False: every iteration)
# The ONE environment switch of the BatchNorm-backward fusions (A/B of the whole feature against stand-alone reduce passes): SEGLAND_BN_FUSE=0 switches all three off.
_BN_FUSE = os.environ.get('SEGLAND_BN_FUSE', '1') != '0'        # BN-backward statistics in the data-gradient epilogues (conv_gemm_common.h: conv_epilogue_fast MODE 3)
_BN_DUAL = _BN_FUSE        # test hook: bn3 + downsample BN backward in one sweep each (bn.hip reduce2 / apply2)
_BN_CROSS = _BN_FUSE       # test hook: bn3's column sums from the NEXT block's conv1 data-gradient epilogue (pixel-stationary kernel MODE 5)


def conv_bn_bwd(dy, y_mask, c, x, conv, bn, mean, invstd, need_dx, need_dw, want_dres=False, addend=None, x2=None, dx_out=None,
                bits=None, addend_bits=None, pre_partial=None, below=None, bn_done=None, prev3=None, prevd=None, dx_half=False, addend_half=False, wbatch=None):
    """Backward of y = act(bn(conv(x))).  Returns (dx, dw, dgamma, dbeta, dres, partial_below).
    ReLU gate of dy: `bits` (bit mask from the forward) or `y_mask` (the activation itself).  `addend` (+ optional
    `addend_bits` gate) is accumulated into dx by the dgrad epilogue.
    pre_partial: dy is ALREADY gated and the BN-backward column sums of this layer were produced by the epilogue that wrote it (no reduce pass).
    below = (bits, c, mean, invstd) of the BatchNorm + ReLU that produced x: when the data gradient of this conv runs on a kernel with the staged
    store phase, its epilogue gates dx with those bits and emits that layer's column sums (partial_below is then not None and dx is gated).
    bn_done = (dc, dgamma, dbeta): the BatchNorm part was already done by the caller (ops.bn_bwd2: two BatchNorms behind one ReLU in one sweep).
    prev3 = (bits, c3, mean, invstd) of the PREVIOUS bottleneck's bn3 + output ReLU: dx (with its addend, which must be gated already) is that block's incoming
    gradient; where the pixel-stationary kernel serves the shape it is gated there and reduced against c3 (partial_below = that block's bn3 column sums).
    wbatch (ops.WgradBatch): a flat (1x1) slab reduce joins the caller's one reduce launch -- dw is filled by wbatch.run()."""
    gw, gg, gb = (grad_dst(conv.weight), grad_dst(bn.weight), grad_dst(bn.bias)) if need_dw else (None, None, None)
    if bn_done is not None:
        (dc, dgamma, dbeta), dres = bn_done, None
    else:
        dc, dres, dgamma, dbeta = ops.bn_bwd(dy, None if (bits is not None or pre_partial is not None) else y_mask, c, mean, invstd, bn.weight, train=bn.training,
                                             want_dres=want_dres, mask=None if pre_partial is not None else bits, sync_world=sync_world(bn), dgamma_out=gg, dbeta_out=gb,
                                             pre_partial=pre_partial)
        dgamma, dbeta = grad_alias(dgamma, gg), grad_alias(dbeta, gb)
    spec = spec_of(conv)
    dx = dw = part_below = None
    if need_dw:
        dw = grad_alias(ops.conv2d_bwd_weight(x, dc, spec, x2=x2, out=gw, defer=wbatch), gw)
    if need_dx and dx_half:
        # a 1x1 stride-2 conv (a stage entry's downsample branch): its data gradient is non-zero at the even positions only -- return the DENSE gradient on the conv's own
        # output grid; the consumer adds it at the even positions (ops.conv2d_bwd_data_addend_half), the zero-filled tensor is never written
        _, wb = prepared(conv.weight, c.dtype)
        dx = ops.conv2d_bwd_data(dc, wb, ConvSpec(spec.cin, spec.cout, 1, 1, 0, 1), dc.shape[1:3])
    elif need_dx and addend_half:
        _, wb = prepared(conv.weight, c.dtype)
        dx, part_below = ops.conv2d_bwd_data_addend_half(dc, wb, spec, x.shape[1:3], addend, prev3)
    elif need_dx:
        _, wb = prepared(conv.weight, c.dtype)
        if below is not None and _BN_FUSE and addend is None and x2 is None and dx_out is None:
            r = ops.conv2d_bwd_data_bnstat(dc, wb, spec, x.shape[1:3], *below)
            if r is not None:
                dx, part_below = r
        if prev3 is not None and addend is not None and addend_bits is None and x2 is None and dx_out is None:
            if prevd is not None:       # the block in front is a stage's first one: bn3 + downsample BatchNorm behind its ReLU, both reduced here (partial_below is a pair)
                r = ops.conv2d_bwd_data_addend_bnstat2(dc, wb, spec, x.shape[1:3], addend, *prev3, *prevd)
                if r is not None:
                    dx, part_below = r[0], (r[1], r[2])
            else:
                r = ops.conv2d_bwd_data_addend_bnstat(dc, wb, spec, x.shape[1:3], addend, *prev3)
                if r is not None:
                    dx, part_below = r
        if dx is None:
            dx = ops.conv2d_bwd_data(dc, wb, spec, x.shape[1:3], addend=addend, addend_mask=addend_bits,
                                     C1=(x.shape[3] if x2 is not None else None), out=dx_out)
    return dx, dw, dgamma, dbeta, dres, part_below


# ------------------------------------------------------------------------------------------------ stem
class StemFn(torch.autograd.Function):
    """conv1 7x7 s2 -> bn1 -> relu -> maxpool 3x3 s2 (networks/backbones/resnet.py:124-125). img: NCHW float."""

    @staticmethod
    def forward(ctx, img, w, gamma, beta, net, dtype):
        bn = net.bn1
        c0, part = ops.stem_conv_fwd(img, w.detach(), dtype, bn.training)
        mean, invstd, scale, shift = _bn_coeffs(bn, part, c0.numel() // 64)
        pooled, idx = ops.stem_bn_relu_pool(c0, scale, shift, want_idx=True)
        ctx.net = net
        ctx.save_for_backward(img, c0, idx, mean, invstd, scale, shift)
        return pooled

    @staticmethod
    @once_differentiable
    def backward(ctx, dp):
        img, c0, idx, mean, invstd, scale, shift = ctx.saved_tensors
        bn = ctx.net.bn1
        gg, gb = grad_dst(bn.weight), grad_dst(bn.bias)
        if _BN_FUSE:          # bn1's reduce pass rides in the pool / ReLU backward, which reads c0 anyway (round 5)
            g0, pp = ops.stem_pool_relu_bwd_bnstat(dp.contiguous(), idx, c0, scale, shift, mean, invstd)
        else:
            g0, pp = ops.stem_pool_relu_bwd(dp.contiguous(), idx, c0, scale, shift), None
        dc0, _, dgamma, dbeta = ops.bn_bwd(g0, None, c0, mean, invstd, bn.weight, train=bn.training, sync_world=sync_world(bn), dgamma_out=gg, dbeta_out=gb, pre_partial=pp)
        dw = ops.stem_conv_bwd_weight(img, dc0) if ctx.needs_input_grad[1] else None
        return None, dw, grad_alias(dgamma, gg), grad_alias(dbeta, gb), None, None


# ------------------------------------------------------------------------------------------------ bottleneck
class _BlockLink:
    """What two consecutive bottlenecks of ONE forward pass hand each other for the cross-block bn3 fusion (_BN_CROSS).  The producer's forward makes it
    (bn3: ReLU bits, c3, mean, invstd of its output BatchNorm; out_ptr / out_shape: the tensor it returned), the consumer's forward picks it up from the
    producer module -- checked against its own input -- and keeps it in its ctx; the consumer's backward leaves the column sums its conv1 data-gradient
    epilogue produced in pre3 (with the address of the gradient tensor they belong to), the producer's backward takes them.  Nothing is read from module
    state at backward time."""
    __slots__ = ('bn3', 'bnd', 'pre3', 'out_ptr', 'out_shape')

    def __init__(self):
        self.bn3 = self.bnd = self.pre3 = self.out_ptr = self.out_shape = None      # bnd: (cd, mean, invstd) of the downsample BatchNorm behind the same ReLU (a stage's first block)


class BottleneckFn(torch.autograd.Function):
    """networks/backbones/resnet.py:57-78 as one kernel chain; x and the result are NHWC."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        bns = [blk.bn1, blk.bn2, blk.bn3] + ([blk.downsample[1]] if blk.downsample is not None else [])
        # the hand-over record of the block that produced x IN THIS FORWARD PASS (a second forward before the first backward makes new records:
        # a backward never sees another pass's ReLU bits or column sums)
        pm = blk.__dict__.get('_sl_prev')
        plink = pm.__dict__.get('_sl_link') if pm is not None else None
        ctx.prev_link = plink if (plink is not None and plink.out_ptr == x.data_ptr() and plink.out_shape == tuple(x.shape)) else None
        blk.__dict__['_sl_link'] = None
        if _frozen(ctx, *bns):
            a1 = conv_bn_infer(x, blk.conv1, blk.bn1, relu=True)
            a2 = conv_bn_infer(a1, blk.conv2, blk.bn2, relu=True)
            res = x if blk.downsample is None else conv_bn_infer(x, blk.downsample[0], blk.downsample[1], relu=False)
            return conv_bn_infer(a2, blk.conv3, blk.bn3, relu=blk.last_relu, residual=res)
        c1, a1, m1, i1, k1 = conv_bn_fwd(x, blk.conv1, blk.bn1, relu=True, want_mask=True)
        c2, a2, m2, i2, k2 = conv_bn_fwd(a1, blk.conv2, blk.bn2, relu=True, want_mask=True)
        if blk.downsample is not None:
            cd, res, md, idd = conv_bn_fwd(x, blk.downsample[0], blk.downsample[1], relu=False)
        else:
            cd, res, md, idd = None, x, None, None
        c3, out, m3, i3, k3 = conv_bn_fwd(a2, blk.conv3, blk.bn3, relu=blk.last_relu, residual=res, want_mask=True)
        ctx.blk = blk
        ctx.has_ds = blk.downsample is not None
        # the next bottleneck's backward produces this block's incoming gradient: it may gate it and reduce it against c3 right there (_BN_CROSS)
        link = ctx.link = blk.__dict__['_sl_link'] = _BlockLink()
        dual_ok = ctx.has_ds and _BN_DUAL and blk.downsample[1].training and not sync_world(blk.bn3)      # the consumer's one-sweep dual backward (ops.bn_bwd2) must apply
        link.bn3 = (k3, c3, m3, i3) if (_BN_CROSS and k3 is not None and blk.bn3.training and (not ctx.has_ds or dual_ok) and any(ctx.needs_input_grad)) else None
        link.bnd = (cd, md, idd) if (link.bn3 is not None and ctx.has_ds) else None
        link.out_ptr, link.out_shape = out.data_ptr(), tuple(out.shape)
        saved = [x, c1, a1, m1, i1, k1, c2, a2, m2, i2, k2, c3, m3, i3]
        if ctx.has_ds:
            saved += [cd, md, idd]
        if k3 is not None:
            saved.append(k3)
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        blk = ctx.blk
        sv = ctx.saved_tensors
        x, c1, a1, m1, i1, k1, c2, a2, m2, i2, k2, c3, m3, i3 = sv[:14]
        k3 = sv[-1] if blk.last_relu else None        # ReLU bits of the block output (gates BOTH the bn3 and the shortcut gradient)
        dout = dout.contiguous()
        need_w = ctx.needs_input_grad[2]            # params are all-or-nothing frozen in this model family
        need_x = ctx.needs_input_grad[0]
        # the data gradients of conv3 and conv2 gate their result with the ReLU bits of the layer below and emit its BN-backward column sums in the epilogue
        # (where the kernel has the staged store phase: layer3 / layer4 at the bench shapes): that layer's reduce pass over (g, c) disappears
        # this block's incoming gradient may have been gated and reduced against c3 by the block behind it (its conv1 data gradient epilogue): the tensor
        # autograd hands over must be exactly the one that epilogue wrote (a second consumer of this block's output would have made autograd sum into a new one)
        link = ctx.link
        pre3, link.pre3, link.bn3, link.bnd = link.pre3, None, None, None
        p3 = pdual = None
        if pre3 is not None and pre3[0] == dout.data_ptr() and pre3[1] == tuple(dout.shape):
            if not ctx.has_ds and not isinstance(pre3[2], tuple):
                p3, k3 = pre3[2], None               # dout is gated already: no bits for bn3, none for the identity shortcut
            elif ctx.has_ds and isinstance(pre3[2], tuple):
                pdual = pre3[2]                      # ... and reduced against c3 AND the downsample BatchNorm's input (round 5: the dual store loop of the block behind)
        done3 = doned = None
        if ctx.has_ds and _BN_DUAL and k3 is not None and blk.bn3.training and blk.downsample[1].training and not sync_world(blk.bn3):
            # bn3 and the downsample BN sit behind the same ReLU: one sweep over dout and its bits for both reduces, one for both applies (ops.bn_bwd2)
            cd, md, idd = sv[14:17]
            bnd = blk.downsample[1]
            g3, b3, gd_, bd_ = (grad_dst(blk.bn3.weight), grad_dst(blk.bn3.bias), grad_dst(bnd.weight), grad_dst(bnd.bias)) if need_w else (None,) * 4
            dc3, dg3_, db3_, dcd, dgd_, dbd_ = ops.bn_bwd2(dout, None if pdual is not None else k3, c3, m3, i3, blk.bn3.weight, cd, md, idd, bnd.weight, (g3, b3), (gd_, bd_),
                                                           pre_partials=pdual)
            done3 = (dc3, grad_alias(dg3_, g3), grad_alias(db3_, b3))
            doned = (dcd, grad_alias(dgd_, gd_), grad_alias(dbd_, bd_))
        # the block in front of this one can take its bn3 column sums from this block's conv1 data gradient only if the shortcut gradient enters that epilogue
        # gated already: either dout arrived gated (p3), or the shortcut is a downsample branch, or -- the start of a chain inside a stage -- bn3's apply pass
        # also writes the gated gradient (one extra write of dout's size, repaid by every block further up the stage)
        plink = ctx.prev_link
        prev3 = plink.bn3 if (plink is not None and need_x and _BN_CROSS) else None
        if prev3 is not None and (prev3[1].shape != x.shape or prev3[1].dtype != x.dtype or not ops.conv2d_bwd_data_addend_bnstat_ok(x, spec_of(blk.conv1))):
            prev3 = None
        prevd = plink.bnd if prev3 is not None else None
        want_dres = prev3 is not None and not ctx.has_ds and k3 is not None and p3 is None and done3 is None
        wbatch = ops.WgradBatch() if (need_w and _WGRAD_BATCH) else None       # the 1x1 convs' slab reduces of this block: one launch at the end (round 6)
        fold = (_BN3_FOLD and p3 is not None and not ctx.has_ds and need_w and blk.bn3.training and blk.bn2.training and not sync_world(blk.bn3) and k2 is not None
                and ops.conv2d_bwd_data_bnstat_folded_ok(a2, spec_of(blk.conv3)))
        if fold:
            # bn3's apply pass folded into conv3's two gradients: dout arrived gated with its column sums; dc3 is never formed
            spec3 = spec_of(blk.conv3)
            g3_, b3_ = grad_dst(blk.bn3.weight), grad_dst(blk.bn3.bias)
            rows3 = c3.numel() // c3.shape[-1]
            cA, cB, cC, dg3, db3 = ops.bn_bwd_coeffs(p3, rows3, blk.bn3.weight, m3, i3, dgamma_out=g3_, dbeta_out=b3_)
            wf3, wb3 = prepared(blk.conv3.weight, c3.dtype)
            gx, gsum = ops.conv2d_bwd_weight_dy2(a2, dout, a2)      # [g | a2]^T a2 and the column sums of both from one pass over a2
            n3 = spec3.cout
            xtx, xsum = gx[n3:], gsum[n3:]
            wext, vbias = ops.bn_fold_weights(wf3, wb3, cA, cB, db3, xsum, rows3)
            dg3, db3 = grad_alias(dg3, g3_), grad_alias(db3, b3_)
            da2, p2 = ops.conv2d_bwd_data_bnstat_folded(dout, a2, wext, vbias, spec3, k2, c2, m2, i2)
            gw3 = grad_dst(blk.conv3.weight)
            dw3 = ops.bn_fold_wgrad(gx[:n3], xtx, xsum, wf3, cA, cB, cC, m3, out=gw3 if gw3 is not None else torch.empty((n3, spec3.cin, 1, 1), dtype=torch.float32, device=a2.device))
            dw3 = grad_alias(dw3, gw3)
            dres = None
        else:
            da2, dw3, dg3, db3, dres, p2 = conv_bn_bwd(dout, None, c3, a2, blk.conv3, blk.bn3, m3, i3, True, need_w, bits=k3, pre_partial=p3,
                                                       below=(k2, c2, m2, i2) if blk.bn2.training else None, bn_done=done3, want_dres=want_dres, wbatch=wbatch)
        da1, dw2, dg2, db2, _, p1 = conv_bn_bwd(da2, None, c2, a1, blk.conv2, blk.bn2, m2, i2, True, need_w, bits=None if p2 is not None else k2, pre_partial=p2,
                                                below=(k1, c1, m1, i1) if blk.bn1.training else None, wbatch=wbatch)
        grads_ds = ()
        if ctx.has_ds:
            cd, md, idd = sv[14:17]
            dsc = blk.downsample[0]
            half = (_DS_HALF and need_x and dsc.kernel_size == (1, 1) and dsc.stride == (2, 2) and dsc.padding == (0, 0) and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
                    and ops.conv2d_bwd_data_addend_half_ok(x, spec_of(blk.conv1)))
            dxd, dwd, dgd, dbd, _, _ = conv_bn_bwd(dout, None, cd, x, dsc, blk.downsample[1], md, idd, need_x, need_w, bits=k3, bn_done=doned, dx_half=half, wbatch=wbatch)
            grads_ds = (dwd, dgd, dbd)
            addend, abits = dxd, None
        elif dres is not None:
            addend, abits = dres, None              # identity shortcut, gated by bn3's apply pass (chain start, see above)
        else:
            addend, abits = dout, k3                # identity shortcut: dout * relu'(out), gated inside the dgrad epilogue (k3 None: dout arrived gated)
        if abits is not None:
            prev3 = None
        dx, dw1, dg1, db1, _, pp = conv_bn_bwd(da1, None, c1, x, blk.conv1, blk.bn1, m1, i1, need_x, need_w,
                                               addend=addend if need_x else None, addend_bits=abits if need_x else None, bits=None if p1 is not None else k1, pre_partial=p1,
                                               prev3=prev3, prevd=prevd, addend_half=ctx.has_ds and half, wbatch=wbatch)
        if wbatch is not None:
            wbatch.run()
        if pp is not None and prev3 is not None:
            plink.pre3 = (dx.data_ptr(), tuple(dx.shape), pp)
        return (dx, None, dw1, dg1, db1, dw2, dg2, db2, dw3, dg3, db3) + grads_ds


def bottleneck_params(blk):
    p = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
         blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
    if blk.downsample is not None:
        p += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
    return p


# ------------------------------------------------------------------------------------------------ pyramid pooling
class PPMFn(torch.autograd.Function):
    """networks/pspnet_pop.py:31-35: 4 x (adaptive pool -> 1x1 -> BN -> ReLU -> bilinear up) (+) feats -> 3x3 -> BN -> ReLU -> 1x1+bias.
    The 4096-channel concat is virtual: the 3x3 conv reads [priors | feats] from two tensors."""

    @staticmethod
    def forward(ctx, x4, dec, *params):
        sizes = dec.sizes
        B, H, W, Cf = x4.shape
        Cs = dec.stages[0][1].out_channels
        pooled = ops.ppm_pool_fwd(x4, sizes)
        stage_act = torch.empty((pooled.shape[0], Cs), dtype=torch.float32, device=x4.device)   # stage path is fp32 (see ppm.hip)
        if _frozen(ctx, dec.bottleneck[1], *[st[2] for st in dec.stages]):
            call, _ = ops.ppm_rows_gemm(pooled, _stage_weights(dec)[0], B, sizes)     # the four stage convs in one grouped GEMM
            off = 0
            for s, st in zip(sizes, dec.stages):
                n = B * s * s
                _, _, scale, shift = _bn_eval_coeffs(st[2])
                ops.bn_act(call[off:off + n], scale, shift, relu=True, out=stage_act[off:off + n])
                off += n
            bt = dec.bottleneck
            if _PPM_FACTORISED:
                # as in the training branch: the prior half of the 3x3 conv contracted on the s x s grids (exact), the x4 half on the MFMA kernel with the
                # gathered prior term entering before the folded BatchNorm -- half the FLOPs of the virtual-concat conv, and a shape the patch kernel serves
                N = bt[0].out_channels
                wq_f, _, wf4, _ = _ppm_weights(bt[0].weight, Cs, len(sizes), x4.dtype)
                q, _ = ops.ppm_rows_gemm(stage_act, wq_f, B, sizes)
                gpri = ops.ppm_fact_gather(q, x4.shape, sizes, N, x4.dtype)
                _, _, scale, shift = _bn_eval_coeffs(bt[1])
                ab = ops.conv2d_affine_fwd(x4, wf4, ConvSpec(Cf, N, 3, 1, 1, 1), scale, shift, relu=True, pre_addend=gpri)
            else:
                priors = ops.ppm_upsample_fwd(stage_act, x4.shape, sizes, x4.dtype)
                ab = conv_bn_infer(priors, bt[0], bt[1], relu=True, x2=x4)
            wf, _ = prepared(bt[3].weight, x4.dtype)
            return ops.conv2d_fwd(ab, wf, spec_of(bt[3]), bias=bt[3].bias.detach())[0]
        # the four stage convs as ONE grouped skinny GEMM over the pyramid rows (16..576 rows per level)
        wst_f, _ = _stage_weights(dec)
        call, part = ops.ppm_rows_gemm(pooled, wst_f, B, sizes, want_stats=any(st[2].training for st in dec.stages))
        cl, ml, il, off, grp = [], [], [], 0, ops.ppm_stat_groups(B, sizes)
        for k, (s, st) in enumerate(zip(sizes, dec.stages)):
            n = B * s * s
            c = call[off:off + n]
            m, i, scale, shift = _bn_coeffs(st[2], part[grp[k]:grp[k + 1]] if st[2].training else None, n)
            ops.bn_act(c, scale, shift, relu=True, out=stage_act[off:off + n])
            cl.append(c); ml.append(m); il.append(i); off += n
        bt = dec.bottleneck
        ctx.fact = _PPM_FACTORISED
        if ctx.fact:
            # prior half of the 3x3 conv contracted on the s x s grids (exact; see ppm.hip), x4 half on the MFMA kernel
            N = bt[0].out_channels
            wq_f, wq_b, wf4, wb4 = _ppm_weights(bt[0].weight, Cs, len(sizes), x4.dtype)
            q, _ = ops.ppm_rows_gemm(stage_act, wq_f, B, sizes)
            gpri = ops.ppm_fact_gather(q, x4.shape, sizes, N, x4.dtype)
            spec4 = ConvSpec(Cf, N, 3, 1, 1, 1)
            cb, part = ops.conv2d_fwd(x4, wf4, spec4, pre_addend=gpri, want_stats=bt[1].training)
            mb, ib, scale, shift = _bn_coeffs(bt[1], part, cb.numel() // N)
            ab, kb = ops.bn_act(cb, scale, shift, relu=True, want_mask=True)
            priors = cb.new_empty(0)
        else:
            priors = ops.ppm_upsample_fwd(stage_act, x4.shape, sizes, x4.dtype)
            cb, ab, mb, ib = conv_bn_fwd(priors, bt[0], bt[1], relu=True, x2=x4)
            kb = cb.new_empty(0, dtype=torch.uint8)
        wf, _ = prepared(bt[3].weight, x4.dtype)
        feat, _ = ops.conv2d_fwd(ab, wf, spec_of(bt[3]), bias=bt[3].bias.detach())
        ctx.dec = dec
        ctx.save_for_backward(x4, pooled, stage_act, priors, cb, ab, mb, ib, *cl, *ml, *il, kb, call)
        return feat

    @staticmethod
    @once_differentiable
    def backward(ctx, dfeat):
        dec = ctx.dec
        sizes, nl = dec.sizes, len(dec.sizes)
        sv = ctx.saved_tensors
        x4, pooled, stage_act, priors, cb, ab, mb, ib = sv[:8]
        cl, ml, il = sv[8:8 + nl], sv[8 + nl:8 + 2 * nl], sv[8 + 2 * nl:8 + 3 * nl]
        kb = sv[8 + 3 * nl]                          # ReLU bits of the bottleneck BatchNorm (factorised path)
        B, H, W, Cf = x4.shape
        Cs = stage_act.shape[1]
        bt = dec.bottleneck
        dfeat = dfeat.contiguous()
        need_w = ctx.needs_input_grad[2]
        need_x = ctx.needs_input_grad[0]
        spec_f = spec_of(bt[3])
        _, wbf = prepared(bt[3].weight, x4.dtype)
        # the classifier conv's data gradient gates its result with the bottleneck ReLU's bits and emits the bottleneck BatchNorm's backward column sums
        # in its epilogue where the kernel serves the shape (round 5: one 36 us reduce pass less per step)
        dab = ppart = None
        if ctx.fact and _BN_FUSE and bt[1].training and kb.numel():
            r = ops.conv2d_bwd_data_bnstat(dfeat, wbf, spec_f, (H, W), kb, cb, mb, ib)
            if r is not None:
                dab, ppart = r
        if dab is None:
            dab = ops.conv2d_bwd_data(dfeat, wbf, spec_f, (H, W))
        gwf = grad_dst(bt[3].weight) if need_w else None
        dwf = dbias = None
        if need_w:
            # weight + bias gradient of the biased 1x1 conv in one kernel (the bias column sums come out of the weight-gradient kernel's dy fragments: round 6, one pass over dfeat less)
            dwf, dbias = ops.conv2d_bwd_weight_bias(ab, dfeat, spec_f, out=gwf)
            dwf = grad_alias(dwf, gwf)
        if ctx.fact:
            N = bt[0].out_channels
            wq_f, wq_b, wf4, wb4 = _ppm_weights(bt[0].weight, Cs, nl, x4.dtype)
            ggb, gbb = (grad_dst(bt[1].weight), grad_dst(bt[1].bias)) if need_w else (None, None)
            dcb, _, dgb, dbb = ops.bn_bwd(dab, None, cb, mb, ib, bt[1].weight, train=bt[1].training, sync_world=sync_world(bt[1]), dgamma_out=ggb, dbeta_out=gbb,
                                          mask=None if ppart is not None else kb, pre_partial=ppart)
            dgb, dbb = grad_alias(dgb, ggb), grad_alias(dbb, gbb)
            spec4 = ConvSpec(Cf, N, 3, 1, 1, 1)
            dcat = ops.conv2d_bwd_data(dcb, wb4, spec4, (H, W))                     # gradient of the x4 half only: [B,H,W,Cf]
            cat_off = 0
            dwb = None
            gq = ops.ppm_fact_scatter(dcb, x4.shape, sizes)
            dstage, _ = ops.ppm_rows_gemm(gq, wq_b, B, sizes)
            if need_w:
                gwb = grad_dst(bt[0].weight)
                dwb = gwb if gwb is not None else torch.empty_like(bt[0].weight, dtype=torch.float32)
                ops.conv2d_bwd_weight(x4, dcb, spec4, out=dwb, out_ci_off=nl * Cs)
                dwq = torch.empty((nl, 9 * N, Cs), dtype=torch.float32, device=x4.device)
                if _PPM_WGRAD_GROUPED:
                    ops.ppm_rows_wgrad(gq, stage_act, B, sizes, outs=[dwq[k] for k in range(nl)])      # all levels in one launch (round 6)
                else:
                    qspec, off = ConvSpec(Cs, 9 * N, 1), 0
                    for k, s in enumerate(sizes):
                        n = B * s * s
                        ops.conv2d_bwd_weight(stage_act[off:off + n].view(B, s, s, Cs), gq[off:off + n].view(B, s, s, 9 * N), qspec,
                                              out=dwq[k].view(9 * N, Cs, 1, 1))
                        off += n
            if need_w:
                ops.ppm_dwq_scatter(dwq, dwb, Cs, nl)
                dwb = grad_alias(dwb, gwb)
        else:
            dcat, dwb, dgb, dbb, _, _ = conv_bn_bwd(dab, ab, cb, priors, bt[0], bt[1], mb, ib, True, need_w, x2=x4)
            dstage = ops.ppm_upsample_bwd(dcat, x4.shape, sizes, Cs)
            cat_off = len(sizes) * Cs
        dc_all = torch.empty_like(stage_act)
        gstage, off = [], 0
        grouped = None
        call = sv[9 + 3 * nl]
        if _BN_FUSE and _STAGE_BN_GROUPED and call.numel() and not any(sync_world(st[2]) for st in dec.stages):
            # all levels' BatchNorm + ReLU backward in ONE launch (round 5): twelve latency-bound launches less per step
            dsts = [(grad_dst(st[2].weight), grad_dst(st[2].bias)) if need_w else (None, None) for st in dec.stages]
            tmp = torch.empty((nl, 2, Cs), dtype=torch.float32, device=x4.device)
            dgl = [d_[0] if d_[0] is not None else tmp[k, 0] for k, d_ in enumerate(dsts)]
            dbl = [d_[1] if d_[1] is not None else tmp[k, 1] for k, d_ in enumerate(dsts)]
            ops.ppm_stage_bn_bwd(dstage, stage_act, call, B, sizes, ml, il, [st[2].weight for st in dec.stages], [st[2].training for st in dec.stages], dgl, dbl, out=dc_all)
            grouped = [(grad_alias(dgl[k], dsts[k][0]), grad_alias(dbl[k], dsts[k][1])) for k in range(nl)]
        dws_all = None
        if need_w and _PPM_WGRAD_GROUPED and grouped is not None:
            # the four stage convs' weight gradients in one launch (ops.ppm_rows_wgrad); dc_all is complete (the grouped BatchNorm backward wrote every level)
            dws_all = ops.ppm_rows_wgrad(dc_all, pooled, B, sizes, outs=[grad_dst(st[1].weight) for st in dec.stages])
        for k, (s, st) in enumerate(zip(sizes, dec.stages)):
            n = B * s * s
            ggs, gbs, gws = (grad_dst(st[2].weight), grad_dst(st[2].bias), grad_dst(st[1].weight)) if need_w else (None, None, None)
            if grouped is not None:
                dgs, dbs = grouped[k]
            else:
                _, _, dgs, dbs = ops.bn_bwd(dstage[off:off + n], stage_act[off:off + n], cl[k], ml[k], il[k], st[2].weight, train=st[2].training,
                                            out=dc_all[off:off + n], sync_world=sync_world(st[2]), dgamma_out=ggs, dbeta_out=gbs)
                dgs, dbs = grad_alias(dgs, ggs), grad_alias(dbs, gbs)
            if dws_all is not None:
                dws = grad_alias(dws_all[k] if dws_all[k] is gws else dws_all[k].view_as(st[1].weight), gws)
            else:
                dws = grad_alias(ops.conv2d_bwd_weight(pooled[off:off + n].view(B, s, s, Cf), dc_all[off:off + n].view(B, s, s, Cs), spec_of(st[1]), out=gws), gws) if need_w else None
            gstage += [dws, dgs, dbs]; off += n
        dpooled = ops.ppm_rows_gemm(dc_all, _stage_weights(dec)[1], B, sizes)[0] if need_x else None
        dx4 = ops.ppm_pool_bwd(dpooled, x4.shape, x4.dtype, sizes, dcat=dcat, cat_off=cat_off) if need_x else None
        return (dx4, None, *gstage, dwb, dgb, dbb, dwf, dbias)


import os as _os
_PPM_FACTORISED = _os.environ.get('SEGLAND_PPM_DIRECT') != '1'


def set_ppm_factorised(flag):
    """Test hook: choose between the factorised prior path (default) and the direct virtual-concat 3x3 conv."""
    global _PPM_FACTORISED
    _PPM_FACTORISED = bool(flag)


def _stage_weights(dec):
    """The stage 1x1 weights for the grouped GEMMs: (per level [Cs][Cf] forward, per level [Cf][Cs] data gradient) -- the float copies weight preparation makes anyway
    (refresh_weights: the stage convs are in the plan with float32), handed to ops.ppm_rows_gemm as per-level pointers.  (Until round 6 they were stacked and transposed
    with two torch launches per step: 37 us.)"""
    fs, bs = [], []
    for st in dec.stages:
        wf, wb = prepared(st[1].weight, torch.float32)
        fs.append(wf); bs.append(wb)
    return fs, bs


def _ppm_weights(w, Cs, nl, dtype):
    """Per-level 1x1 weights of the factorised prior path (float) + GEMM layouts of the x4 channel slice, cached on the Parameter."""
    ent = getattr(w, '_sl_ppm', None)
    if ent is None or ent[0] != _wver(w) or ent[1] != dtype or ent[2] != w.data_ptr():
        wq_f, wq_b = ops.ppm_wq_prep(w, Cs, nl)
        wf4, wb4 = ops.weight_prep_slice(w, dtype, nl * Cs, w.shape[1] - nl * Cs)
        N = w.shape[0]
        ent = (_wver(w), dtype, w.data_ptr(), wq_f, wq_b, wf4, wb4)
        w._sl_ppm = ent
    return ent[3], ent[4], ent[5], ent[6]


def ppm_params(dec):
    p = []
    for st in dec.stages:
        p += [st[1].weight, st[2].weight, st[2].bias]
    bt = dec.bottleneck
    return p + [bt[0].weight, bt[1].weight, bt[1].bias, bt[3].weight, bt[3].bias]


# ------------------------------------------------------------------------------------------------ POP head
def _row_parts(R, big):
    """Row ranges the MLP GEMMs run on: the pixel rows (a multiple of 256: whole 256-row tiles, ONE round of the persistent tile kernel on 256 CUs) and the
    handful of prototype rows behind them as a launch of their own -- in one launch 65 536 + 14 rows are 257 tiles, i.e. a second round for 14 rows (measured: the
    512 -> 512 GEMMs of the head ran at 375 TFLOP/s, half of what the same kernel reaches on 65 536 rows)."""
    if big and 0 < big < R and big % 256 == 0 and big >= 256 * 96:
        return [(0, big), (big, R)]
    return [(0, R)]


def _mlp_fwd(X, cls, big=0):
    """classifier MLP (pspnet_pop.py:46-52) on rows X [R,512]: two MFMA 1x1 convs with fused ReLU, then a row dot.  big: number of leading pixel rows (see _row_parts)."""
    R, Cn = X.shape
    w1f, _ = prepared(cls[0].weight, X.dtype)
    w2f, _ = prepared(cls[2].weight, X.dtype)
    h1, h2 = torch.empty_like(X), torch.empty_like(X)
    for a, b in _row_parts(R, big):
        ops.conv2d_fwd(X[a:b].view(1, 1, b - a, Cn), w1f, spec_of(cls[0]), relu=True, out=h1[a:b].view(1, 1, b - a, Cn))
        ops.conv2d_fwd(h1[a:b].view(1, 1, b - a, Cn), w2f, spec_of(cls[2]), relu=True, out=h2[a:b].view(1, 1, b - a, Cn))
    w3 = cls[4].weight.detach().view(-1)
    z = ops.rowdot_fwd(h2, w3)
    return h1.view(1, 1, R, Cn), h2.view(1, 1, R, Cn), z


def _mlp_bwd(X, h1, h2, cls, dz, need_w, need_x, big=0):
    R, Cn = X.shape
    w3 = cls[4].weight.detach().view(-1)
    dh2, dw3 = ops.rowdot_bwd(h2.view(R, Cn), w3, dz)
    h1r = h1.view(R, Cn)
    _, w2b = prepared(cls[2].weight, X.dtype)
    dh1 = torch.empty_like(dh2)
    parts = _row_parts(R, big)
    for a, b in parts:
        ops.conv2d_bwd_data(dh2[a:b].view(1, 1, b - a, Cn), w2b, spec_of(cls[2]), (1, b - a), mask_src=h1r[a:b].view(1, 1, b - a, Cn), out=dh1[a:b].view(1, 1, b - a, Cn))
    dh2, dh1 = dh2.view(1, 1, R, Cn), dh1.view(1, 1, R, Cn)
    g2 = grad_dst(cls[2].weight) if need_w else None
    dw2 = grad_alias(ops.conv2d_bwd_weight(h1, dh2, spec_of(cls[2]), out=g2), g2) if need_w else None
    dX = None
    if need_x:
        _, w1b = prepared(cls[0].weight, X.dtype)
        dX = torch.empty_like(X)
        for a, b in parts:
            ops.conv2d_bwd_data(dh1[0, 0, a:b].view(1, 1, b - a, Cn), w1b, spec_of(cls[0]), (1, b - a), out=dX[a:b].view(1, 1, b - a, Cn))
    g1 = grad_dst(cls[0].weight) if need_w else None
    dw1 = grad_alias(ops.conv2d_bwd_weight(X.view(1, 1, R, Cn), dh1, spec_of(cls[0]), out=g1), g1) if need_w else None
    return dX, dw1, dw2, (dw3.view_as(cls[4].weight) if need_w else None)


def _base_chain(model, S_b, dtype, frozen):
    """ft mode: the base classifier over the +-base prototype rows (pspnet_pop.py:210-216).  With the base prototypes and the base classifier frozen (ft_pop.py:197-203)
    its result is the same every iteration: kept on the model, keyed on the versions of what it is computed from (four launches per step less).  Never created while a
    graph is being captured (its tensors would live in the graph's pool)."""
    def run():
        Xb = torch.empty((2 * S_b.shape[0], S_b.shape[1]), dtype=dtype, device=S_b.device)
        ops.pop_proto_rows(S_b.contiguous(), Xb)
        return (Xb,) + tuple(_mlp_fwd(Xb, model.classifier))
    key = base_chain_key(model) if frozen else None
    if key is None:
        return run()
    key = key + (dtype, tuple(S_b.shape))
    ent = model.__dict__.get('_sl_base_chain')
    if ent is not None and ent[0] == key:
        return ent[1]
    out = run()
    if not (S_b.is_cuda and torch.cuda.is_current_stream_capturing()):
        model.__dict__['_sl_base_chain'] = (key, out)
    return out


def base_chain_key(model):
    """What the cached rows of _base_chain are computed from (versions + addresses of the frozen base classifier's weights and of base_emb), or None when the cache does
    not apply.  graph_step.GraphedStep puts it into its state key: a captured fine-tune step has the cached tensors baked in, so a change of these weights behind a
    graph (load_state_dict, init_cls_n) must force a re-capture -- every other weight is picked up through its pointer, these rows are not (round-5 advisor)."""
    emb = getattr(model, 'base_emb', None)
    cls = getattr(model, 'classifier', None)
    if not (_BASE_CHAIN_CACHE and getattr(model, 'is_ft', False) and emb is not None and cls is not None and not emb.requires_grad):
        return None
    ws = cls_params(cls)
    if any(w.requires_grad for w in ws):
        return None
    return (tuple((_wver(w), w.data_ptr()) for w in ws), emb._version, emb.data_ptr())


class PopHeadFn(torch.autograd.Function):
    """orthogonal_decompose + classifier(s) in the collapsed form.
    feat NHWC [B,h,w,512]; S_b / S_n: L2-normalised prototypes (float).  Returns preds [B, 1+Kb+Kn, h, w] float,
    channel order [bg | base | novel] (pspnet_pop.py:159,219).
    Base mode (cls_n is None): one MLP chain (classifier) over [bg rows ; +-S_b].
    ft mode: classifier over [+-S_b] (base scalars), classifier_n over [bg rows ; +-S_n]."""

    @staticmethod
    def forward(ctx, feat, S_b, S_n, model, *params):
        B, h, w, Cn = feat.shape
        R, N = B * h * w, h * w
        Kb = S_b.shape[0]
        Kn = 0 if S_n is None else S_n.shape[0]
        ft = S_n is not None
        S = torch.cat([S_b, S_n], 0).contiguous() if ft else S_b.contiguous()
        feats2d = feat.view(R, Cn)
        S_main = S_n.contiguous() if ft else S            # prototypes whose +- rows ride the bg chain
        Km = S_main.shape[0]
        X = torch.empty((R + 2 * Km, Cn), dtype=feat.dtype, device=feat.device)
        proj = ops.pop_decompose_into(feats2d, S, X[:R])
        ops.pop_proto_rows(S_main, X[R:])
        cls_main = model.classifier_n if ft else model.classifier
        h1, h2, z = _mlp_fwd(X, cls_main, big=R)
        if ft:
            Xb, h1b, h2b, zb = _base_chain(model, S_b, feat.dtype, frozen=not (ctx.needs_input_grad[1] or ctx.needs_input_grad[4]))
            a = torch.cat([zb[:Kb], z[R:R + Kn]]).contiguous()
            b = torch.cat([zb[Kb:], z[R + Kn:]]).contiguous()
        else:
            Xb = h1b = h2b = None
            a, b = z[R:R + Kb].contiguous(), z[R + Kb:].contiguous()
        z_bg = z[:R].contiguous()
        preds = ops.pop_combine_fwd(proj, z_bg, a, b, B, N)
        ctx.model, ctx.ft, ctx.dims = model, ft, (B, h, w, Cn, Kb, Kn)
        ctx.save_for_backward(feat, S, X, proj, h1, h2, a, b, *([Xb, h1b, h2b] if ft else []))
        return preds.view(B, 1 + Kb + Kn, h, w)

    @staticmethod
    @once_differentiable
    def backward(ctx, dpreds):
        model, ft = ctx.model, ctx.ft
        B, h, w, Cn, Kb, Kn = ctx.dims
        R, N = B * h * w, h * w
        sv = ctx.saved_tensors
        feat, S, X, proj, h1, h2, a, b = sv[:8]
        ni = ctx.needs_input_grad
        need_feat, need_sb, need_sn = ni[0], ni[1], ni[2]
        dpreds = dpreds.contiguous().view(B, 1 + Kb + Kn, N)
        dz_bg, dproj, da, db = ops.pop_combine_bwd(dpreds, proj, a, b, B, N)
        cls_main = model.classifier_n if ft else model.classifier
        npar = 3
        need_w_main = ni[4 + (npar if ft else 0)]
        if ft:
            dz = torch.cat([dz_bg, da[Kb:], db[Kb:]]).contiguous()
        else:
            dz = torch.cat([dz_bg, da, db]).contiguous()
        dX, dw1, dw2, dw3 = _mlp_bwd(X, h1, h2, cls_main, dz, need_w_main, need_feat or need_sb or need_sn, big=R)
        gb = (None, None, None)
        dS = None
        dfeat = None
        if need_feat or need_sb or need_sn:
            dq, dS = ops.pop_decompose_bwd(dX[:R], feat.view(R, Cn), S, proj, dproj)
            dfeat = dq.view(B, h, w, Cn) if need_feat else None
            Km = Kn if ft else Kb
            rows = dX[R:].float()
            dS_rows = rows[:Km] - rows[Km:]
            if ft:
                dS = torch.cat([dS[:Kb], dS[Kb:] + dS_rows], 0)
            else:
                dS = dS + dS_rows
        if ft:
            Xb, h1b, h2b = sv[8:11]
            need_w_b = ni[4]
            if need_w_b or need_sb:
                dzb = torch.cat([da[:Kb], db[:Kb]]).contiguous()
                dXb, bw1, bw2, bw3 = _mlp_bwd(Xb, h1b, h2b, model.classifier, dzb, need_w_b, need_sb)
                gb = (bw1, bw2, bw3)
                if need_sb:
                    rb = dXb.float()
                    dS = torch.cat([dS[:Kb] + rb[:Kb] - rb[Kb:], dS[Kb:]], 0)
            dSb = dS[:Kb].contiguous() if (need_sb and dS is not None) else None
            dSn = dS[Kb:].contiguous() if (need_sn and dS is not None) else None
            return (dfeat, dSb, dSn, None, *gb, dw1, dw2, dw3)
        return (dfeat, dS if need_sb else None, None, None, dw1, dw2, dw3)


def cls_params(cls):
    return [cls[0].weight, cls[2].weight, cls[4].weight]


class ProtoFn(torch.autograd.Function):
    """F.normalize of the prototype embeddings (pspnet_pop.py:96-99), their similarity matrix (:185-186 / :236-239) and the orthogonality term
    (criterion.py:37-43) as ONE kernel forward and ONE backward (torch: ~50 launches of 5 us).  forward(Ea [Ka,C], Eb [Kb,C] | None) ->
    (Sa, Sb | empty, orth []): rows of the similarity are the `a` prototypes, columns [a ; b]."""

    @staticmethod
    def forward(ctx, Ea, Eb):
        Ea = Ea.detach().float().contiguous()
        Ebc = None if Eb is None else Eb.detach().float().contiguous()
        Sa, Sb, inv, G, orth = ops.pop_proto_fwd(Ea, Ebc)
        ctx.has_b = Eb is not None
        ctx.save_for_backward(Sa, inv, G, *([Sb] if ctx.has_b else []))
        Sb_out = Sb if ctx.has_b else Sa.new_empty(0)
        if not (ctx.has_b and ctx.needs_input_grad[1]):
            # every output of a Function requires grad as soon as ONE input does: without this the frozen base prototypes of ft mode (base_emb.requires_grad False,
            # ft_pop.py:197-203) look trainable to PopHeadFn -- its frozen-base-chain cache never engaged and it ran the base MLP backward for a gradient nobody takes
            ctx.mark_non_differentiable(Sb_out)
        return Sa, Sb_out, orth.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dSa, dSb, dorth):
        sv = ctx.saved_tensors
        Sa, inv, G = sv[:3]
        Sb = sv[3] if ctx.has_b else None
        need_a, need_b = ctx.needs_input_grad[0], ctx.has_b and ctx.needs_input_grad[1]
        if not (need_a or need_b):
            return None, None
        dSa = None if dSa is None else dSa.float().contiguous()
        dSb = None if (dSb is None or not ctx.has_b) else dSb.float().contiguous()
        dorth = None if dorth is None else dorth.float().reshape(1).contiguous()
        dEa, dEb = ops.pop_proto_bwd(Sa, Sb, inv, G, dSa, dSb, dorth, need_a, need_b)
        return dEa, dEb


# ------------------------------------------------------------------------------------------------ loss
class UpsampleCEFn(torch.autograd.Function):
    """F.interpolate(align_corners=True) + CrossEntropyLoss(ignore_index, mean) of loss/criterion.py:51-52, fused."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits = logits.contiguous()
        out = ops.upsample_ce_fwd(logits, target, ignore_index)
        ctx.ignore = ignore_index
        ctx.save_for_backward(logits, target, out)
        return out[0].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, target, out = ctx.saved_tensors
        gs = g.detach().reshape(1).float().contiguous()
        return ops.upsample_ce_bwd(logits, target, out, gs, ctx.ignore), None, None
