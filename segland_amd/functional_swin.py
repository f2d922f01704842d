"""Block-level autograd Functions of the Swin-POP path: each one runs a fused chain of libsegland_hip.so kernels
forward and the hand-written backward chain, exactly as functional.py does for the ResNet path.  nn.Linear / nn.LayerNorm / nn.Conv2d /
nn.BatchNorm2d modules are parameter holders (state_dict compatibility with networks/swin_pop.py); their own forward is never called.

Layout: token maps and decoder maps are NHWC [B,H,W,P] in the compute dtype, P = ops_swin.pad_to(C) (zero channel pad where C = 96).
Linear layers = 1x1 convs on the MFMA implicit-GEMM kernels with zero-padded weight copies; gradients are sliced back to the parameter shapes.
"""
import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import ops
from . import ops_swin as osw
from .functional import _nbt_pending, _wver
from .ops import ConvSpec
from .ops_swin import pad_to

_PSP_GROUPED = True    # test hook: the UperNet pyramid's stage BatchNorm backward + stage weight gradients as two grouped launches (False: a per-level chain)
_GELU_FUSE = True      # test hook: fc2's data gradient lands behind the GELU in its epilogue (sl_conv2d_bwd_data_gelu); False: + a gelu_bwd launch (profiles/r5_ab_swin_gelu.txt)
_BIAS_TAIL = True      # test hook: d qkv.bias = column sums + pad-token share inside the relative-position table launch (False: a torch add)
_WGRAD_BATCH = True    # test hook: the slab reduces of a block's four nn.Linear weight gradients in one launch (ops.WgradBatch); False: one reduce launch behind every weight gradient
_BN_BIAS_ZERO = True   # test hook: the exact-zero bias gradient of a conv in front of a train-mode BatchNorm is written as zero (False: the column sum of the BN-input gradient)
_LN_SCALE = True       # test hook: DropPath's per-sample factor on a branch's incoming gradient comes out of the LayerNorm backward that produced the gradient (False: a scale_add launch)
_RESIZE_ADD = True     # test hook: AddResizedFn as one launch (False: clone + accumulate)
_BN_BIAS_FOLD = True   # test hook: a biased conv's bias enters running_mean inside bn_finalize_train (False: a torch add_ launch per BatchNorm)


# ------------------------------------------------------------------------------------------------ prepared (padded) weights
class _Lin:
    __slots__ = ('wf', 'wb', 'bias', 'spec', 'N', 'K', 'Np', 'Kp', 'k', 'stage')


def _padded(t, shape, fill=0.0):
    if tuple(t.shape) == tuple(shape):
        return t.detach()
    out = torch.full(shape, fill, dtype=torch.float32, device=t.device)
    out[tuple(slice(0, s) for s in t.shape)] = t.detach()
    return out


class _Plan:
    """Every GEMM weight of a Swin-POP model prepared in ONE kernel launch per optimizer step (the ResNet path's functional._PrepPlan for this
    model family): lin_prep registers a weight the first time it sees it; refresh(), called at the top of a forward, re-pads the padded ones into
    their persistent fp32 staging buffers and rebuilds all GEMM layouts with sl_weight_prep_batched."""

    def __init__(self):
        self.items, self.table, self.total, self.sig = {}, None, 0, None
        self.vecs, self.ctable, self.ctotal, self.cn = {}, None, 0, 0      # padded BatchNorm gamma / beta copies (_bn_padded) ride in the same copy launch
        self.rels, self.rtable = {}, None                                  # relative-position bias tiles of the blocks: one gather launch per refresh

    def register(self, weight, bias, dtype, Kp, Np, k, col_map, L, stage):
        self.items[(id(weight), Kp, Np, dtype)] = (weight, bias, dtype, Kp, Np, k, col_map, L, stage)
        self.sig = None

    def register_vec(self, dst, src, on_refresh):
        """dst[:len(src)] <- src (fp32 vectors) with every refresh; on_refresh() tells the owner that its copy is current."""
        self.vecs[(dst.data_ptr(), src.data_ptr())] = (dst, src, on_refresh)
        self.sig = None

    def register_rel(self, attn, out):
        """out [heads, n, n] <- relative_position_bias_table[relative_position_index] with every refresh (see _rel_bias)."""
        self.rels[id(attn)] = (attn, out)
        self.sig = None

    def refresh(self):
        import struct
        if not self.items:
            return
        stale = [it for it in self.items.values() if getattr(it[0], '_sl_lin', {}).get((it[3], it[4], it[2]), (None,))[0] != _lin_key(it[0], it[1], it[2])]
        if not stale:
            return
        if len(stale) * 2 < len(self.items):             # a few trainable weights over a frozen model (ft_pop): individual launches are cheaper
            return
        sig = tuple((id(it[0]), it[0].data_ptr(), it[3], it[4], it[2]) for it in self.items.values()) + tuple(self.vecs.keys()) + \
            tuple((k, a.relative_position_bias_table.data_ptr(), o.data_ptr()) for k, (a, o) in self.rels.items())
        if sig != self.sig:
            rec, start = b'', 0
            for (w, b, dtype, Kp, Np, k, cmap, L, stage) in self.items.values():
                src = stage if stage is not None else w.detach()
                assert Np % 64 == 0 and Kp % 32 == 0 and k * k <= 9
                rec += struct.pack('<QQQiiiiq', src.data_ptr(), L.wf.data_ptr(), L.wb.data_ptr(), Np, Kp, k * k, ops.dt(dtype), start)
                start += Np * Kp // 2048
            dev = next(iter(self.items.values()))[0].device
            self.table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
            self.total = start
            # the strided re-fills of the padded staging buffers (weights, biases, BatchNorm vectors) as ONE launch: table of (dst, src, rows, cols, dst pitch)
            crec, cstart, cn = b'', 0, 0

            def add(dst, src, rows, cols, pitch):
                nonlocal crec, cstart, cn
                crec += struct.pack('<QQiiiiq', dst.data_ptr(), src.data_ptr(), rows, cols, pitch, 0, cstart)
                cstart += (rows * cols + 1023) // 1024
                cn += 1
            for (w, b, dtype, Kp, Np, k, cmap, L, stage) in self.items.values():
                ok = w.dtype == torch.float32 and w.is_contiguous()
                if stage is not None and cmap is None and ok:
                    add(stage, w, w.shape[0], w.shape[1] * k * k, Kp * k * k)
                if L.bias is not None and b is not None and L.bias.data_ptr() != b.data_ptr() and b.dtype == torch.float32 and b.is_contiguous():
                    add(L.bias, b, 1, b.shape[0], L.bias.shape[0])
            for (dst, src, _) in self.vecs.values():
                add(dst, src, 1, src.shape[0], dst.shape[0])
            self.ctable = torch.frombuffer(bytearray(crec), dtype=torch.uint8).to(dev) if cn else None
            rrec = b''
            for (a, o) in self.rels.values():
                t, idx = a.relative_position_bias_table, a.relative_position_index
                assert t.dtype == torch.float32 and t.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
                rrec += struct.pack('<QQQii', o.data_ptr(), t.data_ptr(), idx.data_ptr(), t.shape[1], idx.numel())
            self.rtable = torch.frombuffer(bytearray(rrec), dtype=torch.uint8).to(dev) if rrec else None
            self.ctotal, self.cn, self.sig = cstart, cn, sig
        if self.ctable is not None:
            ops.copy2d_multi(self.ctable, self.cn, self.ctotal)
        for (w, b, dtype, Kp, Np, k, cmap, L, stage) in self.items.values():
            ok = w.dtype == torch.float32 and w.is_contiguous()
            if stage is not None and (cmap is not None or not ok):          # zero pad is persistent; only the real block is rewritten
                w4 = w.detach().reshape(w.shape[0], w.shape[1], k, k)
                if cmap is not None:
                    stage[:w.shape[0]].index_copy_(1, cmap, w4)
                else:
                    stage[:w.shape[0], :w.shape[1]].copy_(w4)
            if L.bias is not None and b is not None and L.bias.data_ptr() != b.data_ptr() and not (b.dtype == torch.float32 and b.is_contiguous()):
                L.bias[:b.shape[0]].copy_(b.detach())
        for (_, _, on_refresh) in self.vecs.values():
            on_refresh()
        ops.weight_prep_batched(self.table, len(self.items), self.total)
        if self.rtable is not None:
            ops.relpos_gather_multi(self.rtable, len(self.rels))
            for (a, o) in self.rels.values():
                t = a.relative_position_bias_table
                a.__dict__['_sl_rel'] = ((_wver(t), t.data_ptr()), o)
        for (w, b, dtype, Kp, Np, k, cmap, L, stage) in self.items.values():
            w._sl_lin[(Kp, Np, dtype)] = (_lin_key(w, b, dtype), L)


CURRENT_PLAN = [None]        # the model whose forward is running registers its weights here (GFSS_Model._features); None: per-weight launches


def model_plan(model):
    plan = model.__dict__.get('_sl_swin_plan')
    if plan is None:
        plan = model.__dict__['_sl_swin_plan'] = _Plan()
    return plan


def _lin_key(weight, bias, dtype):
    return (_wver(weight), _wver(bias) if bias is not None else None, weight.data_ptr())


def lin_prep(weight, bias, dtype, Kp=None, Np=None, k=1, col_map=None):
    """GEMM-layout copies of an nn.Linear weight [N,K] (or conv weight [N,K,k,k]) zero-padded to [Np,Kp], cached on the Parameter until it changes.
    col_map (optional): index tensor scattering the K real input columns into a wider padded input (concat of padded maps)."""
    N, K = weight.shape[0], weight.shape[1]
    Kp = pad_to(K) if Kp is None else Kp
    Np = pad_to(N) if Np is None else Np
    key = _lin_key(weight, bias, dtype)
    cache = getattr(weight, '_sl_lin', None)
    if cache is None:
        cache = {}
        weight._sl_lin = cache
    ent = cache.get((Kp, Np, dtype))
    if ent is None or ent[0] != key:
        w4 = weight.detach().reshape(N, K, k, k) if weight.dim() == 2 else weight.detach()
        padded = (Np, Kp) != (N, K) or col_map is not None
        if ent is not None:
            L, stage = ent[1], ent[1].stage
        else:
            L, stage = _Lin(), None
            if padded:
                stage = torch.zeros((Np, Kp, k, k), dtype=torch.float32, device=weight.device)
            L.wf = torch.empty((Np, k, k, Kp), dtype=dtype, device=weight.device)
            L.wb = torch.empty((Kp, k, k, Np), dtype=dtype, device=weight.device)
            L.bias = None if bias is None else (torch.zeros(Np, dtype=torch.float32, device=weight.device) if Np != N else None)
            L.spec = ConvSpec(Kp, Np, k, 1, k // 2, 1)
            L.N, L.K, L.Np, L.Kp, L.k, L.stage = N, K, Np, Kp, k, stage
        if stage is not None:
            if col_map is not None:
                stage[:N].index_copy_(1, col_map, w4.float())
            else:
                stage[:N, :K].copy_(w4)
        src = stage if stage is not None else w4.float().contiguous()
        import ctypes as C
        from . import _lib
        _lib.check(_lib.lib().sl_weight_prep(ops.dt(dtype), ops._p(src), Np, Kp, k, k, ops._p(L.wf), ops._p(L.wb), ops._s()), 'weight_prep')
        if bias is not None:
            if Np != N:
                L.bias[:N].copy_(bias.detach())
            else:
                L.bias = bias.detach()
        if ent is None and CURRENT_PLAN[0] is not None and weight.dtype == torch.float32 and weight.is_contiguous() and weight.is_leaf:
            CURRENT_PLAN[0].register(weight, bias, dtype, Kp, Np, k, col_map, L, stage)
        cache[(Kp, Np, dtype)] = (key, L)
        ent = cache[(Kp, Np, dtype)]
    return ent[1]


def lin_fwd(x, L, residual=None, x2=None, row_scale=None, want_gelu=False):
    """x [B,H,W,Kp] -> [B,H,W,Np] = row_scale[b] * (x W^T + b) + residual (row_scale: DropPath's per-sample factor);
    want_gelu: (y, GELU(y)) from the same epilogue."""
    if x2 is None:
        return ops.linear_fwd(x, L.wf, L.spec, bias=L.bias, row_scale=row_scale, residual=residual, want_gelu=want_gelu)
    assert row_scale is None and not want_gelu
    return ops.conv2d_fwd(x, L.wf, L.spec, bias=L.bias, pre_addend=residual, x2=x2)[0]


def lin_bwd(x, dy, L, need_dx=True, need_w=True, col_map=None, x2=None, batch=None, gelu_h=None, wbatch=None):
    """(dx, dw in the parameter's shape, dbias) of lin_fwd.  batch (ops.ColsumBatch): dbias is filled by batch.run().
    gelu_h: x = GELU(gelu_h) -- dx is then the gradient wrt gelu_h (the activation's backward in the data gradient's epilogue).
    wbatch (ops.WgradBatch): dw is filled by wbatch.run() -- the slab reduces of a block's linears in one launch; wbatch.run() comes before batch.run()."""
    if need_dx and gelu_h is not None:
        dx = ops.conv2d_bwd_data_gelu(dy, L.wb, L.spec, x.shape[1:3], gelu_h)
    else:
        dx = ops.conv2d_bwd_data(dy, L.wb, L.spec, x.shape[1:3], C1=(x.shape[3] if x2 is not None else None)) if need_dx else None
    dw = db = None
    if need_w:
        clip = col_map is None and x2 is None and L.K < L.spec.cin          # zero-padded input channels: the reduce writes the parameter's shape, no slicing copy
        if wbatch is not None and col_map is None and x2 is None and L.k == 1:
            r = ops.conv2d_bwd_weight_clip(x, dy, L.spec, L.N, L.K, want_bias=L.bias is not None, batch=batch, defer=wbatch)
            dwp, db = r if L.bias is not None else (r, None)
            if db is not None:
                db = db[:L.N]
                db = db.contiguous() if batch is None else db
            return dx, dwp.view(L.N, L.K), db
        if L.bias is not None:
            if clip:
                dwp, db = ops.conv2d_bwd_weight_clip(x, dy, L.spec, L.N, L.K, want_bias=True, batch=batch)
            else:
                dwp, db = ops.conv2d_bwd_weight_bias(x, dy, L.spec, x2=x2, batch=batch)       # bias gradient in the weight gradient's reduce launch
            db = db[:L.N]
            db = db.contiguous() if batch is None else db
        elif clip:
            dwp = ops.conv2d_bwd_weight_clip(x, dy, L.spec, L.N, L.K)
        else:
            dwp = ops.conv2d_bwd_weight(x, dy, L.spec, x2=x2)
        dw = dwp if clip else (dwp[:L.N].index_select(1, col_map) if col_map is not None else dwp[:L.N, :L.K]).contiguous()
        if L.k == 1:
            dw = dw.view(L.N, L.K) if dw.shape[1] == L.K else dw
    return dx, dw, db


def _rel_bias(attn):
    """relative_position_bias_table gathered to [heads, 49, 49] (swintransformer.py:128-131), cached per table version."""
    t = attn.relative_position_bias_table
    key = (_wver(t), t.data_ptr())
    ent = attn.__dict__.get('_sl_rel')
    if ent is None or ent[0] != key:
        n = attn.relative_position_index.shape[0]
        ent = (key, t.detach()[attn.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous().float())
        attn.__dict__['_sl_rel'] = ent
        plan = CURRENT_PLAN[0]
        if plan is not None and t.dtype == torch.float32 and t.is_contiguous() and t.is_leaf and id(attn) not in plan.rels:
            plan.register_rel(attn, ent[1])              # from the next optimizer step on the tile is refilled in place by the plan's one gather launch
    return ent[1]


def _ones(B, dev):
    return torch.ones(B, dtype=torch.float32, device=dev)


# ------------------------------------------------------------------------------------------------ patch embedding
class PatchEmbedFn(torch.autograd.Function):
    """swintransformer.py:413-433: conv 4x4 s4 + LayerNorm -> tokens [B,H/4,W/4,P]."""

    @staticmethod
    def forward(ctx, img, pe, dtype, w, b, gamma, beta):
        Cn = w.shape[0]
        P = pad_to(Cn)
        tok = osw.patch_embed_fwd(img.contiguous(), w.detach().contiguous(), b.detach(), dtype, P)
        if pe.norm is None:
            ctx.has_norm = False
            ctx.save_for_backward(img)
            ctx.Cn = Cn
            return tok
        y, stats = osw.layernorm_fwd(tok, gamma.detach(), beta.detach(), Cn)
        ctx.has_norm, ctx.Cn = True, Cn
        ctx.save_for_backward(img, tok, stats, gamma)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.has_norm:
            img, tok, stats, gamma = ctx.saved_tensors
            dtok, dg, db = osw.layernorm_bwd(dy, tok, gamma.detach(), stats, ctx.Cn)
        else:
            img, = ctx.saved_tensors
            dtok, dg, db = dy, None, None
        dw, dbias = osw.patch_embed_bwd(img.contiguous(), dtok, ctx.Cn)
        return None, None, None, dw, dbias, dg, db


# ------------------------------------------------------------------------------------------------ Swin block
class SwinLink:
    """What two consecutive blocks of ONE forward pass hand each other in the backward: the block behind (it runs first) produces this block's incoming gradient in its
    LayerNorm-1 backward and leaves the copy scaled with this block's MLP DropPath factor s2 here (pre = (address, shape, tensor)) -- this block's scale_add launch disappears."""
    __slots__ = ('s2', 'pre')

    def __init__(self, s2):
        self.s2, self.pre = s2, None


class SwinBlockFn(torch.autograd.Function):
    """swintransformer.py:195-250 on a token map x [B,H,W,P]:
        x1 = x + s1 * proj(W-MSA(qkv(LN1(x))));   out = x1 + s2 * fc2(GELU(fc1(LN2(x1))))
    s1, s2: per-sample DropPath scales (0 or 1/keep, timm) or None.  plink: the SwinLink of the block that produced x (None: not a block, or no DropPath there);
    link: this block's own (None when s2 is None)."""

    @staticmethod
    def forward(ctx, x, blk, s1, s2, plink, link, *params):
        n1w, n1b, table, qw, qb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = params
        a = blk.attn
        Cn, heads, shift = blk.dim, blk.num_heads, blk.shift_size
        dtp = x.dtype
        P = x.shape[-1]
        Lq = lin_prep(qw, qb, dtp, Kp=P, Np=pad_to(3 * Cn))
        Lp = lin_prep(pw, pb, dtp, Kp=P, Np=P)
        L1 = lin_prep(f1w, f1b, dtp, Kp=P)
        L2 = lin_prep(f2w, f2b, dtp, Np=P)
        rel = _rel_bias(a)
        xn, st1 = osw.layernorm_fwd(x, n1w.detach(), n1b.detach(), Cn)
        qkv = lin_fwd(xn, Lq)
        att = osw.window_attention_fwd(qkv, qb.detach().contiguous(), rel, Cn, heads, shift, P)
        x1 = lin_fwd(att, Lp, residual=x, row_scale=s1)            # DropPath scale + shortcut in the GEMM epilogue
        xn2, st2 = osw.layernorm_fwd(x1, n2w.detach(), n2b.detach(), Cn)
        h, g = lin_fwd(xn2, L1, want_gelu=True)                     # pre-activation (for the backward) and GELU from one epilogue
        out = lin_fwd(g, L2, residual=x1, row_scale=s2)
        if any(ctx.needs_input_grad):
            ctx.blk, ctx.plink, ctx.link = blk, plink, link
            # g ([B,H,W,4C], the largest tensor of the block) is kept rather than recomputed: 0.6 GB over a Swin-T at 8 tiles of 512x512
            ctx.save_for_backward(x, st1, xn, qkv, att, x1, st2, xn2, h, g, rel, s1, s2, *params)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        blk = ctx.blk
        sv = ctx.saved_tensors
        x, st1, xn, qkv, att, x1, st2, xn2, h, g, rel, s1, s2 = sv[:13]
        n1w, n1b, table, qw, qb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = sv[13:]
        Cn, heads, shift = blk.dim, blk.num_heads, blk.shift_size
        dtp, P = x.dtype, x.shape[-1]
        need_w = ctx.needs_input_grad[6]               # parameters of a block are frozen or trainable together
        Lq = lin_prep(qw, qb, dtp, Kp=P, Np=pad_to(3 * Cn))
        Lp = lin_prep(pw, pb, dtp, Kp=P, Np=P)
        L1 = lin_prep(f1w, f1b, dtp, Kp=P)
        L2 = lin_prep(f2w, f2b, dtp, Np=P)
        dout = dout.contiguous()
        link, plink = ctx.link, ctx.plink
        pre = None
        if link is not None:
            pre, link.pre = link.pre, None
        if s2 is None:
            dz = dout
        elif pre is not None and pre[0] == dout.data_ptr() and pre[1] == tuple(dout.shape):
            dz = pre[2]                                # the block behind wrote dout AND s2 * dout (autograd handed over exactly that tensor: no second consumer summed into a new one)
        else:
            dz = osw.scale_add(dout, s2)
        # the eight column sums of this backward (four bias gradients, two LayerNorm (dgamma, dbeta) pairs, the attention bias and the pad-token
        # bias) are finalised by ONE launch at the end
        batch = ops.ColsumBatch() if need_w else None
        wbatch = ops.WgradBatch() if (need_w and _WGRAD_BATCH) else None       # the four linears' slab reduces: one launch at the end (round 6)
        if _GELU_FUSE:
            dh, dw2, db2 = lin_bwd(g, dz, L2, need_w=need_w, batch=batch, gelu_h=h, wbatch=wbatch)       # fc2's data gradient * GELU'(h) in one launch
        else:
            dg, dw2, db2 = lin_bwd(g, dz, L2, need_w=need_w, batch=batch, wbatch=wbatch)
            dh = osw.gelu_bwd(h, dg)
        dxn2, dw1, db1 = lin_bwd(xn2, dh, L1, need_w=need_w, batch=batch, wbatch=wbatch)
        if s1 is not None and _LN_SCALE:
            (dx1, dpr), dg2, dbt2 = osw.layernorm_bwd(dxn2, x1, n2w.detach(), st2, Cn, addend=dout, want_param_grads=need_w, batch=batch, row_scale=s1)
        else:
            dx1, dg2, dbt2 = osw.layernorm_bwd(dxn2, x1, n2w.detach(), st2, Cn, addend=dout, want_param_grads=need_w, batch=batch)
            dpr = dx1 if s1 is None else osw.scale_add(dx1, s1)
        datt, dwp, dbp = lin_bwd(att, dpr, Lp, need_w=need_w, batch=batch, wbatch=wbatch)
        dqkv, drel, dpad = osw.window_attention_bwd(qkv, qb.detach().contiguous(), rel, datt, Cn, heads, shift, batch=batch)
        dxn, dwq, dbq = lin_bwd(xn, dqkv, Lq, need_w=need_w, batch=batch, wbatch=wbatch)
        if plink is not None and plink.s2 is not None and _LN_SCALE:
            (dx, dxs), dg1, dbt1 = osw.layernorm_bwd(dxn, x, n1w.detach(), st1, Cn, addend=dx1, want_param_grads=need_w, batch=batch, row_scale=plink.s2)
            plink.pre = (dx.data_ptr(), tuple(dx.shape), dxs)
        else:
            dx, dg1, dbt1 = osw.layernorm_bwd(dxn, x, n1w.detach(), st1, Cn, addend=dx1, want_param_grads=need_w, batch=batch)
        dtable = None
        if need_w:
            if wbatch is not None:
                wbatch.run()
            batch.run()
            # table row t collects the (query, key) pairs with relative offset t: a fixed-order gather instead of index_add_ (atomics,
            # last-bit differences from run to run) -- the step stays bit-reproducible like the rest of the path
            praw = dpad.permute(1, 0, 2)                               # back to the kernel's own [heads, 3, 32]
            if _BIAS_TAIL and praw.is_contiguous() and dbq.is_contiguous() and dbq.numel() == 3 * Cn:
                # the zero-padded tokens' k / v are the bias itself (swintransformer.py:208-213): their share joins the column sums inside the table launch
                dtable, dbq = osw.relpos_table_grad(drel.view(heads, -1), _rel_pairs(blk.attn, table.shape[0]), table.shape[0], qkv_bias=(dbq, praw))
            else:
                dbq = (dbq.view(3, heads, 32) + dpad).view(3 * Cn)
                dtable = osw.relpos_table_grad(drel.view(heads, -1), _rel_pairs(blk.attn, table.shape[0]), table.shape[0])
        return (dx, None, None, None, None, None, dg1, dbt1, dtable, dwq, dbq, dwp, dbp, dg2, dbt2, dw1, db1, dw2, db2)


def _rel_pairs(attn, rows):
    """[rows, m] int32: the flat (query, key) pair indices that read relative-position table row t, -1 padded (m = the largest pair count)."""
    ent = attn.__dict__.get('_sl_relpairs')
    if ent is None or ent.device != attn.relative_position_index.device:
        idx = attn.relative_position_index.view(-1).cpu()
        order = torch.argsort(idx, stable=True)
        counts = torch.bincount(idx, minlength=rows)
        pairs = torch.full((rows, int(counts.max())), -1, dtype=torch.int32)
        start = 0
        for t in range(rows):
            c = int(counts[t])
            pairs[t, :c] = order[start:start + c].to(torch.int32)
            start += c
        ent = attn.__dict__['_sl_relpairs'] = pairs.to(attn.relative_position_index.device).contiguous()
    return ent


def block_params(blk):
    a, m = blk.attn, blk.mlp
    return [blk.norm1.weight, blk.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
            blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]


# ------------------------------------------------------------------------------------------------ patch merging / output norms
class PatchMergeFn(torch.autograd.Function):
    """swintransformer.py:264-290: 2x2 space-to-depth -> LayerNorm(4C) -> Linear(4C, 2C, bias=False)."""

    @staticmethod
    def forward(ctx, x, Cn, rw, gamma, beta):
        xm = osw.merge_gather(x, Cn)
        xn, st = osw.layernorm_fwd(xm, gamma.detach(), beta.detach(), 4 * Cn)
        L = lin_prep(rw, None, x.dtype)
        y = lin_fwd(xn, L)
        ctx.Cn, ctx.xshape = Cn, tuple(x.shape)
        ctx.save_for_backward(xm, st, xn, rw, gamma)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xm, st, xn, rw, gamma = ctx.saved_tensors
        need_w = ctx.needs_input_grad[2]
        L = lin_prep(rw, None, xm.dtype)
        dxn, dw, _ = lin_bwd(xn, dy.contiguous(), L, need_w=need_w)
        dxm, dg, db = osw.layernorm_bwd(dxn, xm, gamma.detach(), st, 4 * ctx.Cn, want_param_grads=need_w)
        return osw.merge_scatter(dxm, ctx.xshape, ctx.Cn), None, dw, dg, db


class LayerNormFn(torch.autograd.Function):
    """norm{i} of swintransformer.py:634-637 (the NCHW permute of :639 is a layout change the NHWC decoder does not need)."""

    @staticmethod
    def forward(ctx, x, Cn, gamma, beta):
        y, st = osw.layernorm_fwd(x, gamma.detach(), beta.detach(), Cn)
        ctx.Cn = Cn
        ctx.save_for_backward(x, st, gamma)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, st, gamma = ctx.saved_tensors
        dx, dg, db = osw.layernorm_bwd(dy.contiguous(), x, gamma.detach(), st, ctx.Cn, want_param_grads=ctx.needs_input_grad[2])
        return dx, None, dg, db


# ------------------------------------------------------------------------------------------------ decoder pieces
def _bn_padded(bn, P):
    """BatchNorm parameter / statistic vectors at the channel pitch (pad channels: gamma 1, beta 0 -> they stay exactly zero).  The running statistics
    LIVE in the padded buffers: bn.running_mean / running_var are re-pointed to views of them, so the finalize kernel updates the module's buffers in
    place (state_dict, load_state_dict and .to() see ordinary [C] tensors; a .to() breaks the alias, which is re-established here).  gamma / beta are
    copied once per optimizer step."""
    Cn = bn.num_features
    if P == Cn:
        return bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var
    ent = bn.__dict__.get('_sl_pad')
    dev = bn.weight.device
    if ent is None or ent['P'] != P or ent['rm'].device != dev or bn.running_mean.data_ptr() != ent['rm'].data_ptr() or bn.running_var.data_ptr() != ent['rv'].data_ptr():
        rm = torch.zeros(P, dtype=torch.float32, device=dev); rv = torch.ones(P, dtype=torch.float32, device=dev)
        with torch.no_grad():
            rm[:Cn].copy_(bn.running_mean); rv[:Cn].copy_(bn.running_var)
        bn.running_mean, bn.running_var = rm[:Cn], rv[:Cn]
        ent = bn.__dict__['_sl_pad'] = {'P': P, 'rm': rm, 'rv': rv, 'gw': torch.ones(P, dtype=torch.float32, device=dev),
                                        'gb': torch.zeros(P, dtype=torch.float32, device=dev), 'key': None}
    key = (_wver(bn.weight), _wver(bn.bias), bn.weight.data_ptr())
    if ent['key'] != key:
        with torch.no_grad():
            ent['gw'][:Cn].copy_(bn.weight); ent['gb'][:Cn].copy_(bn.bias)
        ent['key'] = key
        plan = CURRENT_PLAN[0]
        if plan is not None and not ent.get('planned') and bn.weight.dtype == torch.float32 and bn.weight.is_contiguous() and bn.bias.is_contiguous():
            # from the next optimizer step on these two copies ride in the plan's one copy launch (the key is then current right after refresh())
            def current(ent=ent, bn=bn):
                ent['key'] = (_wver(bn.weight), _wver(bn.bias), bn.weight.data_ptr())
            plan.register_vec(ent['gw'], bn.weight.detach(), current)
            plan.register_vec(ent['gb'], bn.bias.detach(), current)
            ent['planned'] = True
    return ent['gw'], ent['gb'], ent['rm'], ent['rv']


def _bn_forward(bn, c, part, conv_bias, P):
    """Train: batch statistics of the raw conv output c (a conv bias shifts the mean only: it cancels in the normalised output and is added to
    running_mean); eval: running statistics with the bias folded into the shift.  Returns (mean, invstd, scale, shift)."""
    Cn = bn.num_features
    gw, gb, rm, rv = _bn_padded(bn, P)
    if bn.training:
        count = c.numel() // P
        if count <= 1:
            raise ValueError('Expected more than 1 value per channel when training, got %d' % count)
        cb = conv_bias.detach() if (conv_bias is not None and _BN_BIAS_FOLD and conv_bias.dtype == torch.float32 and conv_bias.is_contiguous()) else None
        mean, invstd, scale, shift = ops.bn_finalize_train(part, count, gw.contiguous(), gb.contiguous(), rm, rv, bn.momentum, bn.eps, conv_bias=cb)
        if conv_bias is not None and cb is None:
            with torch.no_grad():
                bn.running_mean.add_(conv_bias.detach(), alpha=bn.momentum)
        _nbt_pending.append(bn.num_batches_tracked)
        bn.__dict__['_sl_rs_epoch'] = bn.__dict__.get('_sl_rs_epoch', 0) + 1
        return mean, invstd, scale, shift
    mean, invstd, scale, shift = ops.bn_finalize_eval(gw.contiguous(), gb.contiguous(), rm.contiguous(), rv.contiguous(), bn.eps)
    if conv_bias is not None:
        bp = _padded(conv_bias, (P,))
        shift = shift + bp * scale
        mean = mean - bp                    # the backward normalises the RAW conv output c: (c + b - running_mean) = c - (running_mean - b)
    return mean, invstd, scale, shift


class ConvBnReluFn(torch.autograd.Function):
    """nn.Sequential(Conv2d(k x k, bias), BatchNorm2d, ReLU) of swin_pop.py:112-131 on an NHWC map [B,H,W,Pin] -> [B,H,W,Pout]."""

    @staticmethod
    def forward(ctx, x, seq, w, b, gamma, beta):
        conv, bn = seq[0], seq[1]
        L = lin_prep(w, None, x.dtype, Kp=x.shape[-1], k=conv.kernel_size[0])
        c, part = ops.conv2d_fwd(x, L.wf, L.spec, want_stats=bn.training)
        mean, invstd, scale, shift = _bn_forward(bn, c, part, b, L.Np)
        y, bits = ops.bn_act(c, scale, shift, relu=True, want_mask=True)
        if any(ctx.needs_input_grad):
            ctx.seq = seq
            ctx.save_for_backward(x, c, mean, invstd, bits, w, gamma)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        conv, bn = ctx.seq[0], ctx.seq[1]
        x, c, mean, invstd, bits, w, gamma = ctx.saved_tensors
        L = lin_prep(w, None, x.dtype, Kp=x.shape[-1], k=conv.kernel_size[0])
        need_w = ctx.needs_input_grad[2]
        Cn = bn.num_features
        gw = _bn_padded(bn, L.Np)[0]
        dc, _, dgam, dbet = ops.bn_bwd(dy.contiguous(), None, c, mean, invstd, gw, train=bn.training, mask=bits)
        dx, dw, _ = lin_bwd(x, dc, L, need_dx=ctx.needs_input_grad[0], need_w=need_w)
        dbias = None
        if need_w and bn.training and _BN_BIAS_ZERO:
            # a conv bias in front of a train-mode BatchNorm: the gradient is EXACTLY zero (the normalisation removes the mean; the column sum of the BN-input gradient is
            # rounding noise -- 2e-2 of the weight gradient's scale in bf16, 5e-6 in the reference's fp32): written as zero instead of summed (two launches less per layer)
            dbias = torch.zeros(Cn, dtype=torch.float32, device=dc.device)
        elif need_w:
            dbias = ops.colsum_rows(dc)[:Cn].contiguous()
        return dx, None, dw, dbias, (dgam[:Cn].contiguous() if need_w else None), (dbet[:Cn].contiguous() if need_w else None)


class ResizeFn(torch.autograd.Function):
    """F.interpolate(size, mode='bilinear', align_corners) / nn.Upsample on an NHWC map."""

    @staticmethod
    def forward(ctx, x, size, align):
        ctx.hw, ctx.align = tuple(x.shape[1:3]), align
        return osw.bilinear_fwd(x, size, align)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return osw.bilinear_bwd(dy.contiguous(), ctx.hw, ctx.align), None, None


class AddResizedFn(torch.autograd.Function):
    """base + F.interpolate(x, base.shape, align_corners=True)   (top-down path swin_pop.py:150-153; sum of the level heads :167-169)."""

    @staticmethod
    def forward(ctx, base, x, align):
        ctx.hw, ctx.align, ctx.same = tuple(x.shape[1:3]), align, tuple(x.shape[1:3]) == tuple(base.shape[1:3])
        if ctx.same:
            return osw.scale_add(x, _ones(x.shape[0], x.device), base)
        if _RESIZE_ADD and base.is_contiguous():
            return osw.bilinear_fwd(x, tuple(base.shape[1:3]), align, base=base)      # base + resize(x) in one pass (round 6: no clone of base first)
        out = base.clone()
        return osw.bilinear_fwd(x, tuple(base.shape[1:3]), align, out=out, accumulate=True)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        return dy, (dy if ctx.same else osw.bilinear_bwd(dy, ctx.hw, ctx.align)), None


class PspSwinFn(torch.autograd.Function):
    """PSPModule of swin_pop.py:7-35: 4 x (adaptive pool -> 1x1 -> BN -> ReLU -> bilinear(align_corners=True)) (+) feats -> 1x1 -> BN -> ReLU
    -> Dropout2d(0.1).  x [B,h,w,Cf]; the concat is virtual ([priors | feats] read from two tensors); the pyramid rows stay fp32."""

    @staticmethod
    def forward(ctx, x, psp, drop, *params):
        sizes, nl = psp.sizes, len(psp.sizes)
        B, h, w, Cf = x.shape
        Cs = psp.stages[0][1].out_channels
        Ps = pad_to(Cs)
        pooled = ops.ppm_pool_fwd(x, sizes)                                                   # [rows, Cf] float
        wkey = tuple((_wver(st[1].weight), st[1].weight.data_ptr()) for st in psp.stages)
        went = psp.__dict__.get('_sl_wst')
        if went is None or went[0] != wkey:
            # the padded stack lives in one buffer (pad rows stay zero); a new optimizer step refills it with ONE multi-tensor copy (was: zeros + copy per level + stack)
            buf = went[1] if (went is not None and went[1].shape == (nl, Ps, Cf) and went[1].device == x.device) else torch.zeros((nl, Ps, Cf), dtype=torch.float32, device=x.device)
            with torch.no_grad():
                torch._foreach_copy_([buf[k, :Cs] for k in range(nl)], [st[1].weight.detach().view(Cs, Cf) for st in psp.stages])
            went = psp.__dict__['_sl_wst'] = (wkey, buf)
        wst = went[1]
        call, part = ops.ppm_rows_gemm(pooled, wst, B, sizes, want_stats=any(st[2].training for st in psp.stages))
        stage_act = torch.empty_like(call)
        priors = torch.empty((B, h, w, nl * Ps), dtype=x.dtype, device=x.device)
        cl, ml, il, off, grp = [], [], [], 0, ops.ppm_stat_groups(B, sizes)
        for k, (s, st) in enumerate(zip(sizes, psp.stages)):
            n = B * s * s
            c = call[off:off + n]
            m, i, scale, shift = _bn_forward(st[2], c, part[grp[k]:grp[k + 1]] if st[2].training else None, None, Ps)
            ops.bn_act(c, scale, shift, relu=True, out=stage_act[off:off + n])
            osw.bilinear_fwd(stage_act[off:off + n].view(B, s, s, Ps), (h, w), True, out=priors, out_off=k * Ps)
            cl.append(c); ml.append(m); il.append(i); off += n
        bt = psp.bottleneck
        wb_, bnb = bt[0].weight, bt[1]
        cmap = _psp_colmap(Cs, Ps, nl, Cf, x.device)
        L = lin_prep(wb_, None, x.dtype, Kp=nl * Ps + Cf, col_map=cmap)
        cb, partb = ops.conv2d_fwd(priors, L.wf, L.spec, x2=x, want_stats=bnb.training)
        mb, ib, scale, shift = _bn_forward(bnb, cb, partb, None, L.Np)
        y, bits = ops.bn_act(cb, scale, shift, relu=True, want_mask=True)
        if drop is not None:
            y = osw.scale_add(y, drop, None, per_channel=True, Cn=bnb.num_features)
        if any(ctx.needs_input_grad):
            ctx.psp = psp
            ctx.save_for_backward(x, pooled, stage_act, priors, cb, mb, ib, bits, drop, wst, *cl, *ml, *il, call)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        psp = ctx.psp
        sizes, nl = psp.sizes, len(psp.sizes)
        sv = ctx.saved_tensors
        x, pooled, stage_act, priors, cb, mb, ib, bits, drop, wst = sv[:10]
        cl, ml, il = sv[10:10 + nl], sv[10 + nl:10 + 2 * nl], sv[10 + 2 * nl:10 + 3 * nl]
        B, h, w, Cf = x.shape
        Cs = psp.stages[0][1].out_channels
        Ps = pad_to(Cs)
        need_w, need_x = ctx.needs_input_grad[3], ctx.needs_input_grad[0]
        bt = psp.bottleneck
        bnb = bt[1]
        dy = dy.contiguous()
        if drop is not None:
            dy = osw.scale_add(dy, drop, None, per_channel=True, Cn=bnb.num_features)
        cmap = _psp_colmap(Cs, Ps, nl, Cf, x.device)
        L = lin_prep(bt[0].weight, None, x.dtype, Kp=nl * Ps + Cf, col_map=cmap)
        gw = _bn_padded(bnb, L.Np)[0]
        dcb, _, dgb, dbb = ops.bn_bwd(dy, None, cb, mb, ib, gw, train=bnb.training, mask=bits)
        dcat, dwb, _ = lin_bwd(priors, dcb, L, need_w=need_w, col_map=cmap, x2=x)
        if need_w:
            dwb = dwb.view(bt[0].weight.shape)
        dstage = torch.empty_like(stage_act)
        dc_all = torch.empty_like(stage_act)
        gstage, off = [], 0
        call = sv[10 + 3 * nl]
        if _PSP_GROUPED:
            # round 6: the four levels' BatchNorm + ReLU backward in one launch and their stage-conv weight gradients in one launch (the PSPNet-POP pyramid's kernels:
            # ops.ppm_stage_bn_bwd, ops.ppm_rows_wgrad) instead of four (reduce, finalize, apply, weight gradient, slab reduce) chains: 16 launches less per step
            for k, s in enumerate(sizes):
                n = B * s * s
                osw.bilinear_bwd(dcat, (s, s), True, out=dstage[off:off + n].view(B, s, s, Ps), Cn=Ps, dy_off=k * Ps)
                off += n
            dgb_l = torch.empty((nl, 2, Ps), dtype=torch.float32, device=x.device)
            ops.ppm_stage_bn_bwd(dstage, stage_act, call, B, sizes, ml, il, [_bn_padded(st[2], Ps)[0] for st in psp.stages], [st[2].training for st in psp.stages],
                                 [dgb_l[k, 0] for k in range(nl)], [dgb_l[k, 1] for k in range(nl)], out=dc_all)
            dws_l = ops.ppm_rows_wgrad(dc_all, pooled, B, sizes) if need_w else None
            for k in range(nl):
                gstage += [dws_l[k].view(Ps, Cf)[:Cs].view_as(psp.stages[k][1].weight) if need_w else None,
                           dgb_l[k, 0, :Cs] if need_w else None, dgb_l[k, 1, :Cs] if need_w else None]
        else:
            for k, (s, st) in enumerate(zip(sizes, psp.stages)):
                n = B * s * s
                osw.bilinear_bwd(dcat, (s, s), True, out=dstage[off:off + n].view(B, s, s, Ps), Cn=Ps, dy_off=k * Ps)
                gwk = _bn_padded(st[2], Ps)[0]
                _, _, dgs, dbs = ops.bn_bwd(dstage[off:off + n], stage_act[off:off + n], cl[k], ml[k], il[k], gwk, train=st[2].training, out=dc_all[off:off + n])
                dws = None
                if need_w:
                    dws = ops.conv2d_bwd_weight(pooled[off:off + n].view(B, s, s, Cf), dc_all[off:off + n].view(B, s, s, Ps), ConvSpec(Cf, Ps, 1))[:Cs].contiguous()
                gstage += [dws, dgs[:Cs].contiguous() if need_w else None, dbs[:Cs].contiguous() if need_w else None]
                off += n
        dx = None
        if need_x:
            dpooled = ops.ppm_rows_gemm(dc_all, wst.transpose(1, 2).contiguous(), B, sizes)[0]
            dx = ops.ppm_pool_bwd(dpooled, x.shape, x.dtype, sizes, dcat=dcat, cat_off=nl * Ps)
        Cb = bnb.num_features
        return (dx, None, None, *gstage, dwb, dgb[:Cb].contiguous() if need_w else None, dbb[:Cb].contiguous() if need_w else None)


_colmaps = {}


def _psp_colmap(Cs, Ps, nl, Cf, dev):
    """Input column of the padded concat [stage_0 (Ps) | ... | stage_{nl-1} (Ps) | feats (Cf)] for every column of the reference's
    [stage_0 (Cs) | ... | feats] bottleneck weight (swin_pop.py:18,34)."""
    key = (Cs, Ps, nl, Cf, dev)
    m = _colmaps.get(key)
    if m is None:
        idx = [l * Ps + c for l in range(nl) for c in range(Cs)] + [nl * Ps + c for c in range(Cf)]
        m = _colmaps[key] = torch.tensor(idx, dtype=torch.long, device=dev)
    return m


def psp_params(psp):
    p = []
    for st in psp.stages:
        p += [st[1].weight, st[2].weight, st[2].bias]
    bt = psp.bottleneck
    return p + [bt[0].weight, bt[1].weight, bt[1].bias]
