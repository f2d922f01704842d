"""Tensor-level wrappers over the C ABI (include/segland_hip.h).

PyTorch supplies device memory (caching allocator) and the current HIP stream only; every arithmetic op on the hot
path is a kernel of libsegland_hip.so.  Activations are NHWC tensors ([B,H,W,C], contiguous) in the compute dtype
(torch.bfloat16 or torch.float32); statistics, logits and weight gradients are float32.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import SL_BF16, SL_F32, SlConvDesc, SlPpmDesc, check

_DT = {torch.float32: SL_F32, torch.bfloat16: SL_BF16}


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise RuntimeError('segland_amd: unsupported compute dtype %s (bfloat16 or float32)' % (t,))


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('segland_amd: tensor is not on the GPU; the HIP path has no CPU fallback')
    if not t.is_contiguous():
        raise RuntimeError('segland_amd: non-contiguous tensor passed to a kernel')
    return t.data_ptr()                  # a plain int: every pointer parameter is declared c_void_p (_lib.declared_functions), ctypes converts


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device


def _s():
    """The current HIP stream of the current device as the C ABI's stream handle.  torch.cuda.current_stream() builds a Stream object (and
    resolves the device through three Python layers): 9 us per launch, ~15 % of the host time of a step; the raw getter is one C call."""
    if _raw_stream is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _f32(n, dev, zero=False):
    return (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=dev)


EPI_STATS, EPI_AFFINE, EPI_ADDEND, EPI_ADDEND_BITS, EPI_GATE, EPI_SPLITK, EPI_GELU = 1, 2, 4, 8, 16, 32, 64      # SL_EPI_* of include/segland_hip.h


# --------------------------------------------------------------------------------------------- live kernel timing
class _Profiler:
    """HIP-event timing of the conv launches on the launch stream (bench.py: roofline of the dominant kernel).
    Off by default.  Launches are attributed to the kernel they are dispatched to (`family`, the name rocprofv3 shows);
    `only` restricts timing to one family so the timed region carries just those event pairs."""

    PEAK_TFLOPS = {SL_BF16: 2500.0, SL_F32: 157.3}      # dense MFMA peaks (MI355X_MICROARCH.md); the floor of a launch is priced against these
    HBM_TBS = 6.3                                       # achievable HBM rate (same guide: 8 TB/s peak, ~6.3 achievable)

    FAMILY = {8: 'conv_gemm_p9_kernel', 7: 'conv_c64k3_kernel', 6: 'conv_gemm_sk_kernel', 5: 'conv_gemm_p8_kernel', 4: 'conv_gemm_ring_kernel', 3: 'conv_rows_small_kernel', 2: 'conv_gemm_glds_kernel', 1: 'conv_gemm_kernel'}

    def __init__(self):
        self.on, self.only, self.rec, self.rec_bytes = False, None, {}, {}

    def start(self, only=None):
        self.on, self.only, self.rec, self.rec_bytes = True, only, {}, {}

    def family(self, kind, d, epi=0):
        if kind == 'conv_wgrad':
            # the kernel rocprofv3 names.  Its deterministic slab reduce (a separate small launch behind it on the same stream) is inside the span
            cfg = _lib.lib().sl_conv2d_wgrad_config(C.byref(d))
            dts = 'bf16' if d.dtype == SL_BF16 else 'f32'
            if cfg == 1:
                return 'conv_wgrad_c64k3_kernel'
            if cfg == 2:
                return 'conv_wgrad_c64p_kernel'
            if cfg == 3:
                return 'conv_wgrad3_kernel'
            name = 'conv_wgrad_glds_kernel' if cfg // 10000000 == 1 else 'conv_wgrad_kernel'
            return '%s<%s, %d, %d>%s' % (name, dts, (cfg % 500000) // 1000, cfg % 1000, ' pixel pairs' if (cfg // 500000) % 2 else '')
        # epi: the SL_EPI_* bits of the launch (the dispatch depends on the epilogue, include/segland_hip.h)
        cfg = _lib.lib().sl_conv2d_tile_config_ex(C.byref(d), 0 if kind == 'conv_fwd' else 1, epi)
        return '%s<%s, %d, %d>%s' % (self.FAMILY.get((cfg // 1000000) % 10, '?'), 'bf16' if d.dtype == SL_BF16 else 'f32', (cfg // 1000) % 1000, cfg % 1000,
                                     ' split-K' if cfg >= 10000000 else '')

    def begin(self, kind, d, epi=0):
        if not self.on:
            return None
        fam = self.family(kind, d, epi)
        if self.only is not None and fam != self.only:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return fam, kind, e0, e1, d

    def end(self, tok, extra_bytes=0):
        """extra_bytes: operand bytes of the launch beyond input + output activations (the residual addend and its gate bits in a data gradient)."""
        if tok is None:
            return
        fam, kind, e0, e1, d = tok
        e1.record()
        gflop = 2.0 * d.B * d.Ho * d.Wo * d.Cout * d.Cin * d.KH * d.KW / 1e9
        es = 2 if d.dtype == SL_BF16 else 4
        gbytes = ((d.B * d.H * d.W * d.Cin + d.B * d.Ho * d.Wo * d.Cout + d.Cout * d.Cin * d.KH * d.KW) * es + extra_bytes) / 1e9
        shape = '%s B%d %dx%d %d->%d k%d s%d d%d' % (kind, d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.stride, d.dil)
        # speed of light of THIS launch: the larger of its MFMA time at the dense peak and its algorithmic bytes at the achievable HBM rate (ms)
        floor = max(gflop / self.PEAK_TFLOPS[d.dtype], gbytes / self.HBM_TBS)
        self.rec.setdefault(fam, []).append((e0, e1, gflop, shape, gbytes, floor))

    def region(self, family, nbytes):
        """Context manager form of begin_bytes / end_bytes (bench.py: the optimizer part of a step)."""
        prof = self

        class _R:
            def __enter__(self_):
                self_.tok = prof.begin_bytes(family, nbytes)

            def __exit__(self_, *exc):
                prof.end_bytes(self_.tok)
        return _R()

    def begin_bytes(self, family, nbytes):
        """HBM-bound kernel families (BatchNorm passes): algorithmic bytes instead of FLOPs; only in the instrumented (un-timed) step."""
        if not self.on or self.only is not None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return family, nbytes, e0, e1

    def end_bytes(self, tok):
        if tok is None:
            return
        family, nbytes, e0, e1 = tok
        e1.record()
        self.rec_bytes.setdefault(family, []).append((e0, e1, nbytes))

    def stop_bytes(self):
        """{family: {calls, ms_total, gbytes}} of the spans recorded since start(); call after stop() (which synchronises)."""
        out = {}
        for fam, evs in self.rec_bytes.items():
            gb = sum(n for _, _, n in evs) / 1e9
            out[fam] = {'calls': len(evs), 'ms_total': sum(a.elapsed_time(b) for a, b, _ in evs), 'gbytes': gb, 'floor_ms': gb / self.HBM_TBS}
        self.rec_bytes = {}
        return out

    def stop(self):
        """{family: {calls, ms_total, gflop, gbytes, floor_ms, shapes: {shape: [calls, ms, gflop, floor_ms]}}}"""
        self.on = False
        torch.cuda.synchronize()
        out = {}
        for fam, evs in self.rec.items():
            ent = out[fam] = {'family': fam, 'calls': 0, 'ms_total': 0.0, 'gflop': 0.0, 'gbytes': 0.0, 'floor_ms': 0.0, 'shapes': {}}
            for e0, e1, gflop, shape, gbytes, floor in evs:
                ms = e0.elapsed_time(e1)
                ent['calls'] += 1; ent['ms_total'] += ms; ent['gflop'] += gflop; ent['gbytes'] += gbytes; ent['floor_ms'] += floor
                sh = ent['shapes'].setdefault(shape, [0, 0.0, 0.0, 0.0])
                sh[0] += 1; sh[1] += ms; sh[2] += gflop; sh[3] += floor
        self.rec = {}
        return out


PROFILER = _Profiler()


# --------------------------------------------------------------------------------------------- convolution
class ConvSpec:
    __slots__ = ('cin', 'cout', 'k', 'stride', 'pad', 'dil')

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1):
        self.cin, self.cout, self.k, self.stride, self.pad, self.dil = cin, cout, k, stride, pad, dil

    def out_hw(self, H, W):
        f = lambda n: (n + 2 * self.pad - self.dil * (self.k - 1) - 1) // self.stride + 1
        return f(H), f(W)


_desc_cache = {}


def conv_desc(dtype, B, H, W, spec, C1=None):
    key = (dtype, B, H, W, spec.cin, spec.cout, spec.k, spec.stride, spec.pad, spec.dil, C1)
    d = _desc_cache.get(key)
    if d is None:
        Ho, Wo = spec.out_hw(H, W)
        d = SlConvDesc(_DT[dtype], B, H, W, spec.cin, spec.cout, spec.k, spec.k, spec.stride, spec.pad, spec.dil, Ho, Wo,
                       spec.cin if C1 is None else C1)
        _desc_cache[key] = d
    return d


def weight_prep(w, dtype, want_fwd=True, want_bwd=True):
    """OIHW float master weight -> (w_fwd [O][kh][kw][I], w_bwd [I][kh][kw][O]) in the compute dtype."""
    O, I, KH, KW = w.shape
    wf = torch.empty((O, KH, KW, I), dtype=dtype, device=w.device) if want_fwd else None
    wb = torch.empty((I, KH, KW, O), dtype=dtype, device=w.device) if want_bwd else None
    wd = w.detach()
    if wd.dtype != torch.float32 or not wd.is_contiguous():
        wd = wd.float().contiguous()
    check(_lib.lib().sl_weight_prep(_DT[dtype], _p(wd), O, I, KH, KW, _p(wf), _p(wb), _s()), 'weight_prep')
    return wf, wb


def weight_prep_batched(table, n, total, nbytes=0):
    """nbytes: algorithmic bytes of the launch (fp32 masters in, both GEMM layouts out), for the profiler only."""
    tok = PROFILER.begin_bytes('weight_prep_batched', nbytes)
    check(_lib.lib().sl_weight_prep_batched(_p(table), n, int(total), _s()), 'weight_prep_batched')
    PROFILER.end_bytes(tok)


def conv2d_fwd(x, wf, spec, x2=None, bias=None, relu=False, want_stats=False, pre_addend=None, out=None):
    B, H, W, C1 = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, C1 if x2 is not None else None)
    y = out if out is not None else torch.empty((B, d.Ho, d.Wo, spec.cout), dtype=x.dtype, device=x.device)
    assert y.numel() == B * d.Ho * d.Wo * spec.cout and y.dtype == x.dtype
    part = None
    if want_stats:
        part = _f32((_lib.lib().sl_conv2d_stat_rows(C.byref(d)), 2, spec.cout), x.device)
    tok = PROFILER.begin('conv_fwd', d, (EPI_STATS if want_stats else 0) | (EPI_AFFINE if (bias is not None or relu) else 0))
    check(_lib.lib().sl_conv2d_fwd_ex(C.byref(d), _p(x), _p(x2), _p(wf), _p(pre_addend), _p(bias), int(relu), _p(y), _p(part), _s()), 'conv2d_fwd')
    PROFILER.end(tok)
    return y, part


def linear_fwd(x, wf, spec, bias=None, row_scale=None, residual=None, want_gelu=False):
    """nn.Linear over an NHWC token map with the block's elementwise tail in the GEMM epilogue:
    y = row_scale[b] * (x w^T + bias) + residual; want_gelu: also returns GELU(y) (evaluated on the stored, rounded y)."""
    B, H, W, _ = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, None)
    y = torch.empty((B, d.Ho, d.Wo, spec.cout), dtype=x.dtype, device=x.device)
    g = torch.empty_like(y) if want_gelu else None
    tok = PROFILER.begin('conv_fwd', d, EPI_AFFINE)
    check(_lib.lib().sl_linear_fwd(C.byref(d), _p(x), _p(wf), _p(bias), _p(row_scale), _p(residual), _p(y), _p(g), _s()), 'linear_fwd')
    PROFILER.end(tok)
    return (y, g) if want_gelu else y


def conv2d_affine_fwd(x, wf, spec, scale, shift, x2=None, residual=None, relu=True, out=None, pre_addend=None):
    """y = act((conv(x|x2) + pre_addend) * scale + shift (+ residual)): eval-mode BN folded into the conv epilogue.  Layers with too few 256-row tiles for the chip
    (a fine-tune pair: 8 192 pixel rows) run split along K through a workspace (sl_conv2d_affine_fwd_workspace says when)."""
    B, H, W, C1 = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, C1 if x2 is not None else None)
    y = out if out is not None else torch.empty((B, d.Ho, d.Wo, spec.cout), dtype=x.dtype, device=x.device)
    L = _lib.lib()
    need = L.sl_conv2d_affine_fwd_workspace(C.byref(d)) if x2 is None else 0
    ws = workspace(need, x.device, 'splitk') if need else None
    tok = PROFILER.begin('conv_fwd', d, EPI_AFFINE | (EPI_SPLITK if need else 0))
    check(L.sl_conv2d_affine_fwd_ex(C.byref(d), _p(x), _p(x2), _p(wf), _p(pre_addend), _p(scale), _p(shift), _p(residual), int(relu), _p(y),
                                    _p(ws), ws.numel() if ws is not None else 0, _s()), 'conv2d_affine_fwd')
    PROFILER.end(tok)
    return y


def conv2d_bwd_data(dy, wb, spec, in_hw, addend=None, mask_src=None, C1=None, out=None, addend_mask=None):
    B = dy.shape[0]
    H, W = in_hw
    d = conv_desc(dy.dtype, B, H, W, spec, C1)
    dx = out if out is not None else torch.empty((B, H, W, spec.cin), dtype=dy.dtype, device=dy.device)
    assert dx.numel() == B * H * W * spec.cin and dx.dtype == dy.dtype
    tok = PROFILER.begin('conv_dgrad', d, (EPI_ADDEND if addend is not None else 0) | (EPI_ADDEND_BITS if addend_mask is not None else 0))
    check(_lib.lib().sl_conv2d_bwd_data(C.byref(d), _p(dy), _p(wb), _p(addend), _p(addend_mask), _p(mask_src), _p(dx), _s()), 'conv2d_bwd_data')
    if tok is not None:
        PROFILER.end(tok, sum(t.numel() * t.element_size() for t in (addend, addend_mask, mask_src) if t is not None))
    return dx


def conv2d_bwd_data_gelu(dy, wb, spec, in_hw, h):
    """conv2d_bwd_data(dy) * GELU'(h) in one launch (h: the stored pre-activation, the data gradient's shape): sl_conv2d_bwd_data_gelu."""
    B = dy.shape[0]
    H, W = in_hw
    d = conv_desc(dy.dtype, B, H, W, spec, None)
    assert (d.Ho, d.Wo) == tuple(dy.shape[1:3]) and tuple(h.shape) == (B, H, W, spec.cin) and h.dtype == dy.dtype and h.is_contiguous()
    dx = torch.empty((B, H, W, spec.cin), dtype=dy.dtype, device=dy.device)
    tok = PROFILER.begin('conv_dgrad', d, EPI_GELU)
    check(_lib.lib().sl_conv2d_bwd_data_gelu(C.byref(d), _p(dy), _p(wb), _p(h), _p(dx), _s()), 'conv2d_bwd_data_gelu')
    PROFILER.end(tok, extra_bytes=h.numel() * h.element_size())
    return dx


def conv2d_bwd_data_bnstat(dy, wb, spec, in_hw, gate, bn_x, mean, invstd):
    """Data gradient gated with the ReLU bits `gate` of its own positions + the reduce pass of the BatchNorm backward below it in the epilogue
    (csrc/conv_gemm_common.h: conv_epilogue_fast MODE 3).  -> (g, partial [rows][2][Cin]) or None when the shape is not served (caller: conv2d_bwd_data + bn_bwd)."""
    B = dy.shape[0]
    H, W = in_hw
    d = conv_desc(dy.dtype, B, H, W, spec, None)
    L = _lib.lib()
    rows = L.sl_conv2d_bwd_data_bnstat_rows(C.byref(d))
    if rows <= 0:
        return None
    dx = torch.empty((B, H, W, spec.cin), dtype=dy.dtype, device=dy.device)
    part = _f32((rows, 2, spec.cin), dy.device)
    tok = PROFILER.begin('conv_dgrad', d, EPI_GATE)
    check(L.sl_conv2d_bwd_data_bnstat(C.byref(d), _p(dy), _p(wb), _p(gate), _p(bn_x), _p(mean), _p(invstd), _p(dx), _p(part), _s()), 'conv2d_bwd_data_bnstat')
    if tok is not None:
        PROFILER.end(tok, bn_x.numel() * bn_x.element_size() + gate.numel())
    return dx, part


def conv2d_bwd_data_addend_bnstat_ok(x, spec):
    """Would conv2d_bwd_data_addend_bnstat serve the data gradient of conv `spec` on input x [B,H,W,Cin]?"""
    if x.dtype != torch.bfloat16:
        return False
    B, H, W, _ = x.shape
    return _lib.lib().sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(conv_desc(x.dtype, B, H, W, spec, None))) > 0


def conv2d_bwd_data_addend_bnstat(dy, wb, spec, in_hw, addend, gate, bn_x, mean, invstd):
    """dx = data gradient + addend, gated with the ReLU bits `gate` of the block output it is the gradient of, + that block's bn3 backward column
    sums (csrc/conv_gemm_sk.hip: pixel-stationary kernel MODE 5).  -> (g, partial [rows][2][Cin]) or None when the shape is not served."""
    B = dy.shape[0]
    H, W = in_hw
    d = conv_desc(dy.dtype, B, H, W, spec, None)
    L = _lib.lib()
    rows = L.sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(d)) if dy.dtype == torch.bfloat16 else 0
    if rows <= 0:
        return None
    dx = torch.empty((B, H, W, spec.cin), dtype=dy.dtype, device=dy.device)
    part = _f32((rows, 2, spec.cin), dy.device)
    tok = PROFILER.begin('conv_dgrad', d, EPI_GATE | EPI_ADDEND)
    check(L.sl_conv2d_bwd_data_addend_bnstat(C.byref(d), _p(dy), _p(wb), _p(addend), _p(gate), _p(bn_x), _p(mean), _p(invstd), _p(dx), _p(part), _s()),
          'conv2d_bwd_data_addend_bnstat')
    if tok is not None:
        PROFILER.end(tok, addend.numel() * addend.element_size() + bn_x.numel() * bn_x.element_size() + gate.numel())
    return dx, part


def conv2d_bwd_data_addend_half_ok(x, spec):
    """Would conv2d_bwd_data_addend_half serve the data gradient of conv `spec` on input x [B,H,W,Cin]?"""
    if x.dtype != torch.bfloat16:
        return False
    B, H, W, _ = x.shape
    return _lib.lib().sl_conv2d_bwd_data_addend_half_ok(C.byref(conv_desc(x.dtype, B, H, W, spec, None))) > 0


def conv2d_bwd_data_addend_half(dy, wb, spec, in_hw, addend_half, prev3=None):
    """dx = data gradient + addend_half [B,H/2,W/2,Cin] at the even positions (the dense data gradient of a 1x1 stride-2 conv on its own grid; its zero-filled
    full-resolution form never exists).  prev3 = (gate bits, bn_x, mean, invstd): also the cross-block statistics of conv2d_bwd_data_addend_bnstat.
    -> (dx, partial or None)."""
    B = dy.shape[0]
    H, W = in_hw
    d = conv_desc(dy.dtype, B, H, W, spec, None)
    L = _lib.lib()
    assert addend_half.shape == (B, H // 2, W // 2, spec.cin) and addend_half.is_contiguous()
    dx = torch.empty((B, H, W, spec.cin), dtype=dy.dtype, device=dy.device)
    part = None
    gate = bn_x = mean = invstd = None
    if prev3 is not None:
        rows = L.sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(d))
        if rows > 0:
            gate, bn_x, mean, invstd = prev3
            part = _f32((rows, 2, spec.cin), dy.device)
    tok = PROFILER.begin('conv_dgrad', d, EPI_ADDEND | (EPI_GATE if part is not None else 0))
    check(L.sl_conv2d_bwd_data_addend_half(C.byref(d), _p(dy), _p(wb), _p(addend_half), _p(gate), _p(bn_x), _p(mean), _p(invstd), _p(dx), _p(part), _s()),
          'conv2d_bwd_data_addend_half')
    if tok is not None:
        PROFILER.end(tok, addend_half.numel() * addend_half.element_size() + (bn_x.numel() * bn_x.element_size() + gate.numel() if part is not None else 0))
    return dx, part


def conv2d_bwd_data_addend_bnstat2(dy, wb, spec, in_hw, addend, gate, bn_x, mean, invstd, bn_x2, mean2, invstd2):
    """The dual form of conv2d_bwd_data_addend_bnstat: the gated result is reduced against the inputs of TWO BatchNorms behind the same ReLU (bn3 + the downsample
    BatchNorm of a stage's first bottleneck).  -> (g, partial, partial2) or None when the shape is not served."""
    B = dy.shape[0]
    H, W = in_hw
    d = conv_desc(dy.dtype, B, H, W, spec, None)
    L = _lib.lib()
    rows = L.sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(d)) if dy.dtype == torch.bfloat16 else 0
    if rows <= 0:
        return None
    dx = torch.empty((B, H, W, spec.cin), dtype=dy.dtype, device=dy.device)
    part = _f32((2, rows, 2, spec.cin), dy.device)
    tok = PROFILER.begin('conv_dgrad', d, EPI_GATE | EPI_ADDEND)
    check(L.sl_conv2d_bwd_data_addend_bnstat2(C.byref(d), _p(dy), _p(wb), _p(addend), _p(gate), _p(bn_x), _p(mean), _p(invstd), _p(bn_x2), _p(mean2), _p(invstd2), _p(dx),
                                              _p(part[0]), _p(part[1]), _s()), 'conv2d_bwd_data_addend_bnstat2')
    if tok is not None:
        PROFILER.end(tok, addend.numel() * addend.element_size() + 2 * bn_x.numel() * bn_x.element_size() + gate.numel())
    return dx, part[0], part[1]


_ws_cache = {}


def after_failed_capture():
    """Drain the device and reset the runtime's sticky last error (csrc/api.cpp sl_hip_clear_error): the kernel-by-kernel step that follows a failed graph capture
    must not have its first launch check report the capture's error."""
    torch.cuda.synchronize()
    _lib.lib().sl_hip_clear_error()


def workspace(nbytes, dev, tag=None):
    """Scratch buffer for ONE call (nothing in it outlives the call's kernels).  Eager: grow-only per device and stream (stream-ordered reuse: all our launches are
    on the current stream); `tag`: a separate buffer (the weight-gradient slabs keep theirs: up to 1 GiB, and the small users do not grow with them).
    While the stream is CAPTURING the cache is not used at all: the buffer comes from the capturing graph's own memory pool and goes back to it when the call returns
    (the pool hands the same block to the next call; kernels of one graph run in capture order).  A cached buffer is not the graph's to keep: every capture of the
    process runs on torch's one capture stream, so the cache entry of that stream was shared by all graphs, and when it grew in the middle of a capture the nodes
    recorded before kept the address of the tensor that was dropped -- memory of an older, already destroyed graph's pool in gpurun r4k..r4m: a GPU memory fault at
    the first replay (tests/test_round4_gpu.py::test_workspace_of_a_captured_call_belongs_to_its_graph)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
    key = (dev, _s(), tag)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = w
    return w


def conv2d_bwd_weight(x, dy, spec, x2=None, out=None, out_ci_off=0, defer=None):
    """dw (float OIHW).  `out`: a wider [Cout][Ctot][k][k] gradient tensor; this conv's channels land at input-channel offset out_ci_off.  The fixed-order slab
    reduce follows the MFMA kernel on the same stream -- or, with defer (WgradBatch), joins the batch's one reduce launch where the shape has a flat (1x1) reduce:
    dw is then filled by defer.run()."""
    B, H, W, C1 = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, C1 if x2 is not None else None)
    dw = out if out is not None else torch.empty((spec.cout, spec.cin, spec.k, spec.k), dtype=torch.float32, device=x.device)
    tot = dw.shape[1]
    if defer is not None and out_ci_off == 0 and tot == spec.cin and dw.is_contiguous():
        L = _lib.lib()
        ws = workspace(L.sl_conv2d_bwd_weight_workspace(C.byref(d)), x.device, 'wgrad_d%d' % len(defer.items))
        item = _lib.SlWgradReduce()
        tok = PROFILER.begin('conv_wgrad', d)
        check(L.sl_conv2d_bwd_weight_defer(C.byref(d), _p(x), _p(x2), _p(dy), _p(dw), 0, 0, _p(ws), ws.numel(), None, C.byref(item), _s()), 'conv2d_bwd_weight_defer')
        PROFILER.end(tok)
        if item.splits > 0:
            defer.items.append(item)
            defer.keep.append((ws, dy, dw))
        return dw
    ws = workspace(_lib.lib().sl_conv2d_bwd_weight_workspace(C.byref(d)), x.device, 'wgrad')
    tok = PROFILER.begin('conv_wgrad', d)
    check(_lib.lib().sl_conv2d_bwd_weight_ex(C.byref(d), _p(x), _p(x2), _p(dy), _p(dw), tot, out_ci_off, _p(ws), ws.numel(), _s()), 'conv2d_bwd_weight')
    PROFILER.end(tok)
    return dw


def _bias_rows(d, n_valid, c_valid):
    """Rows of the bias-gradient partial buffer; the library's own message instead of torch's negative-dimension error when the shape is refused."""
    r = _lib.lib().sl_conv2d_bwd_weight_bias_rows(C.byref(d), n_valid, c_valid)
    if r <= 0:
        check(r if r < 0 else -1, 'conv2d_bwd_weight_bias_rows')
    return r


def conv2d_bwd_weight_bias(x, dy, spec, x2=None, batch=None, out=None):
    """(dw float OIHW, db): weight and bias gradient of a biased 1x1 conv / nn.Linear; the column sums of dy ride in the weight-gradient kernel (or its slab-reduce launch).
    batch (ColsumBatch): db is filled by batch.run().  out: the weight gradient's destination (same shape, float, contiguous)."""
    B, H, W, C1 = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, C1 if x2 is not None else None)
    L = _lib.lib()
    ws = workspace(L.sl_conv2d_bwd_weight_workspace(C.byref(d)), x.device, 'wgrad')
    dw = out if out is not None else torch.empty((spec.cout, spec.cin, spec.k, spec.k), dtype=torch.float32, device=x.device)
    assert dw.numel() == spec.cout * spec.cin * spec.k * spec.k and dw.dtype == torch.float32 and dw.is_contiguous()
    Cn = dy.shape[-1]
    rows = dy.numel() // Cn
    assert Cn == spec.cout and dy.is_contiguous()
    part = _f32((_bias_rows(d, 0, 0), Cn), dy.device)
    tok = PROFILER.begin('conv_wgrad', d)
    check(L.sl_conv2d_bwd_weight_bias(C.byref(d), _p(x), _p(x2), _p(dy), _p(dw), _p(ws), ws.numel(), _p(part), _s()), 'conv2d_bwd_weight_bias')
    PROFILER.end(tok)
    return dw, (batch.add(part) if batch is not None else colsum(part).contiguous())


def conv2d_bwd_weight_clip(x, dy, spec, n_valid, c_valid, x2=None, want_bias=False, batch=None, defer=None):
    """dw in the PARAMETER's shape [n_valid][c_valid][k][k] of a layer computed at zero-padded channel counts (spec.cout x spec.cin): the slab reduce writes only the
    channels that exist.  want_bias: also the bias gradient (padded length spec.cout; batch as in conv2d_bwd_weight_bias)."""
    B, H, W, C1 = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, C1 if x2 is not None else None)
    L = _lib.lib()
    ws = workspace(L.sl_conv2d_bwd_weight_workspace(C.byref(d)), x.device, 'wgrad' if defer is None else 'wgrad_d%d' % len(defer.items))
    dw = torch.empty((n_valid, c_valid, spec.k, spec.k), dtype=torch.float32, device=x.device)
    part = None
    if want_bias:
        Cn = dy.shape[-1]
        assert Cn == spec.cout and dy.is_contiguous()
        part = _f32((_bias_rows(d, n_valid, c_valid), Cn), dy.device)
    tok = PROFILER.begin('conv_wgrad', d)
    if defer is not None:          # defer (WgradBatch): the slab reduce (a flat 1x1 one) joins the batch's one launch; dw (and the bias partials) are filled by defer.run()
        item = _lib.SlWgradReduce()
        check(L.sl_conv2d_bwd_weight_defer(C.byref(d), _p(x), _p(x2), _p(dy), _p(dw), n_valid, c_valid, _p(ws), ws.numel(), _p(part), C.byref(item), _s()), 'conv2d_bwd_weight_defer')
        if item.splits > 0:
            defer.items.append(item)
            defer.keep.append((ws, dy, dw, part))
    else:
        check(L.sl_conv2d_bwd_weight_clip(C.byref(d), _p(x), _p(x2), _p(dy), _p(dw), n_valid, c_valid, _p(ws), ws.numel(), _p(part), _s()), 'conv2d_bwd_weight_clip')
    PROFILER.end(tok)
    if not want_bias:
        return dw
    return dw, (batch.add(part) if batch is not None else colsum(part).contiguous())


class WgradBatch:
    """The slab reduces of several weight gradients in ONE launch (sl_wgrad_reduce_multi): conv2d_bwd_weight_clip(..., defer=batch) runs the split-K kernel and hands out
    dw right away; run() fills all of them (and the bias column-sum partials that ride in the reduce launch: run() comes BEFORE the ColsumBatch's).  Every deferred
    layer keeps a workspace of its own until run()."""

    def __init__(self):
        self.items, self.keep = [], []

    def run(self):
        L = _lib.lib()
        for i0 in range(0, len(self.items), _lib.SL_WGRAD_BATCH_MAX):
            chunk = self.items[i0:i0 + _lib.SL_WGRAD_BATCH_MAX]
            arr = (_lib.SlWgradReduce * len(chunk))(*chunk)
            check(L.sl_wgrad_reduce_multi(arr, len(chunk), _s()), 'wgrad_reduce_multi')
        self.items, self.keep = [], []


# --------------------------------------------------------------------------------------------- batch norm
def bn_finalize_train(part, count, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5, conv_bias=None):
    """conv_bias: the conv in front has a bias that is NOT in the statistics (they are of the raw output): it enters running_mean only."""
    Cn = part.shape[-1]
    o = _f32((4, Cn), part.device)
    if conv_bias is not None:
        assert conv_bias.dtype == torch.float32 and conv_bias.is_contiguous() and conv_bias.numel() <= Cn
        check(_lib.lib().sl_bn_finalize_train_bias(_p(part), part.shape[0], Cn, int(count), _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps,
                                                   _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(conv_bias), conv_bias.numel(), _s()), 'bn_finalize_train_bias')
        return o[0], o[1], o[2], o[3]
    check(_lib.lib().sl_bn_finalize_train(_p(part), part.shape[0], Cn, int(count), _p(gamma), _p(beta), _p(rmean), _p(rvar),
                                          momentum, eps, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _s()), 'bn_finalize_train')
    return o[0], o[1], o[2], o[3]          # mean, invstd, scale, shift


def bn_finalize_eval(gamma, beta, rmean, rvar, eps=1e-5):
    Cn = rmean.shape[0]
    o = _f32((4, Cn), rmean.device)
    check(_lib.lib().sl_bn_finalize_eval(Cn, _p(gamma), _p(beta), _p(rmean), _p(rvar), eps, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _s()),
          'bn_finalize_eval')
    return o[0], o[1], o[2], o[3]


def bn_act(x, scale, shift, residual=None, relu=True, out=None, want_mask=False):
    """y = act(x*scale + shift (+residual)).  want_mask: also return the ReLU bit mask (1 byte per 16-byte vector of y)."""
    Cn = x.shape[-1]
    y = out if out is not None else torch.empty_like(x)
    assert y.numel() == x.numel() and y.dtype == x.dtype
    mask = torch.empty(x.numel() * x.element_size() // 16, dtype=torch.uint8, device=x.device) if (want_mask and relu) else None
    tok = PROFILER.begin_bytes('bn_act_fwd', x.numel() * x.element_size() * (2 + (residual is not None)) + (mask.numel() if mask is not None else 0))
    check(_lib.lib().sl_bn_act_fwd(dt(x), _p(x), _p(scale), _p(shift), _p(residual), int(relu), _p(y), _p(mask), x.numel() // Cn, Cn, _s()), 'bn_act_fwd')
    PROFILER.end_bytes(tok)
    return (y, mask) if want_mask else y


def bn_bwd(dy, y, x, mean, invstd, gamma, train=True, want_dres=False, mask=None, out=None, sync_world=0, dgamma_out=None, dbeta_out=None,
           pre_partial=None):
    """Returns (dx, dres, dgamma, dbeta).  ReLU gate of dy: `mask` (bit mask from bn_act) if given, else y > 0 if y is given.
    sync_world > 0 (SyncBatchNorm semantics, torch/nn/modules/_functions.py): the two column sums that enter dx are all-reduced over
    the process group and the element count is the global one; dgamma / dbeta stay local (DDP averages them like any gradient).
    pre_partial: the (sum g, sum g * xhat) partials already produced by the data-gradient epilogue that wrote dy (conv2d_bwd_data_bnstat: dy is
    gated already, no reduce pass runs here)."""
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    L = _lib.lib()
    nb = x.numel() * x.element_size()
    gate = (mask.numel() if mask is not None else (nb if y is not None else 0))
    if pre_partial is not None:
        assert mask is None and y is None
        part, nblk = pre_partial, pre_partial.shape[0]
    else:
        nblk = L.sl_bn_bwd_reduce_rows(rows, Cn)
        part = _f32((nblk, 2, Cn), x.device)
        tok = PROFILER.begin_bytes('bn_bwd_reduce', 2 * nb + gate)
        check(L.sl_bn_bwd_reduce(dt(x), _p(dy), _p(y), _p(mask), _p(x), _p(mean), _p(invstd), _p(part), rows, Cn, _s()), 'bn_bwd_reduce')
        PROFILER.end_bytes(tok)
    o = _f32((5, Cn), x.device)
    # dgamma_out / dbeta_out: write the parameter gradients straight into the caller's buffers (DDP bucket views, functional.grad_dst)
    dg = o[0] if dgamma_out is None else dgamma_out
    db = o[1] if dbeta_out is None else dbeta_out
    check(L.sl_bn_bwd_finalize(_p(part), nblk, Cn, rows, _p(gamma), _p(mean), _p(invstd), int(train), _p(dg), _p(db), _p(o[2]), _p(o[3]), _p(o[4]), _s()),
          'bn_bwd_finalize')
    if sync_world and train:
        tot = allreduce_partials(part)                      # [2 (hi, lo)][2][C] global sums
        og = _f32((5, Cn), x.device)
        check(L.sl_bn_bwd_finalize(_p(tot), 2, Cn, rows * sync_world, _p(gamma), _p(mean), _p(invstd), 1, _p(og[0]), _p(og[1]), _p(og[2]), _p(og[3]),
                                   _p(og[4]), _s()), 'bn_bwd_finalize(sync)')
        o = torch.stack([o[0], o[1], og[2], og[3], og[4]])  # local dgamma/dbeta, global dx coefficients
    dx = out if out is not None else torch.empty_like(x)
    assert dx.numel() == x.numel() and dx.dtype == x.dtype
    dres = torch.empty_like(x) if want_dres else None
    tok = PROFILER.begin_bytes('bn_bwd_apply', (3 + (dres is not None)) * nb + gate)
    check(L.sl_bn_bwd_apply(dt(x), _p(dy), _p(y), _p(mask), _p(x), _p(o[2]), _p(o[3]), _p(o[4]), _p(mean), _p(dx), _p(dres), rows, Cn, _s()), 'bn_bwd_apply')
    PROFILER.end_bytes(tok)
    return dx, dres, dg, db


def bn_bwd_coeffs(part, rows, gamma, mean, invstd, dgamma_out=None, dbeta_out=None):
    """Train-mode BatchNorm backward up to the coefficients: (cA, cB, cC, dgamma, dbeta) with dx = cA dy + cB (x - mean) + cC, from the column-sum partials a data-gradient
    epilogue produced (no reduce pass, no apply pass: the caller folds the apply into the next kernels, bn_fold_weights / bn_fold_wgrad)."""
    Cn = part.shape[-1]
    o = _f32((5, Cn), part.device)
    dg = o[0] if dgamma_out is None else dgamma_out
    db = o[1] if dbeta_out is None else dbeta_out
    check(_lib.lib().sl_bn_bwd_finalize(_p(part), part.shape[0], Cn, int(rows), _p(gamma), _p(mean), _p(invstd), 1, _p(dg), _p(db), _p(o[2]), _p(o[3]), _p(o[4]), _s()),
          'bn_bwd_finalize')
    return o[2], o[3], o[4], dg, db


def conv2d_bwd_data_bnstat_folded_ok(x, spec):
    """Would conv2d_bwd_data_bnstat_folded serve the data gradient of the 1x1 conv `spec` on input x [B,H,W,Cin]?"""
    if x.dtype != torch.bfloat16 or spec.k != 1:
        return False
    B, H, W, _ = x.shape
    return _lib.lib().sl_conv2d_bwd_data_bnstat_folded_rows(C.byref(conv_desc(x.dtype, B, H, W, spec, None))) > 0


def bn_fold_weights(wf, wb, cA, cB, gsum, xsum, rows):
    """(wt_ext [Cin][Cout + Cin] bf16, bias [Cin]) of conv2d_bwd_data_bnstat_folded from the layer's prepared weights wf [Cout][Cin], wb [Cin][Cout], the BatchNorm-backward
    coefficients of its output (bn_bwd_coeffs) and the column sums of the two inputs (gsum = dbeta, xsum = colsum(x))."""
    Cout, Cin = cA.numel(), xsum.numel()
    assert wf.numel() == Cout * Cin == wb.numel() and wf.dtype == torch.bfloat16 and gsum.numel() == Cout and gsum.is_contiguous() and xsum.is_contiguous()
    wext = torch.empty((Cin, Cout + Cin), dtype=torch.bfloat16, device=wf.device)
    bias = _f32((Cin,), wf.device)
    check(_lib.lib().sl_bn_fold_weights(Cout, Cin, _p(wf), _p(wb), _p(cA), _p(cB), _p(gsum), _p(xsum), int(rows), _p(wext), _p(bias), _s()), 'bn_fold_weights')
    return wext, bias


def bn_fold_wgrad(gtx, xtx, xsum, wf, cA, cB, cC, mean, out=None):
    """The weight gradient [Cout][Cin] of a 1x1 conv whose output BatchNorm's apply pass was folded, from gtx = g^T x, xtx = x^T x, xsum = colsum(x); out: its destination
    (default: in place on gtx)."""
    Cout, Cin = cA.numel(), xsum.numel()
    dw = gtx if out is None else out
    assert gtx.numel() == Cout * Cin == dw.numel() and gtx.is_contiguous() and dw.is_contiguous() and xtx.numel() == Cin * Cin and xtx.is_contiguous()
    assert gtx.dtype == torch.float32 == xtx.dtype == dw.dtype and xsum.is_contiguous()
    check(_lib.lib().sl_bn_fold_wgrad(Cout, Cin, _p(gtx), _p(dw), _p(xtx), _p(xsum), _p(wf), _p(cA), _p(cB), _p(cC), _p(mean), _s()), 'bn_fold_wgrad')
    return dw


def conv2d_bwd_weight_dy2(x, dy1, dy2):
    """[dy1 | dy2]^T x [C1 + C2][Cin] (float) and the column sums of [dy1 | dy2] from ONE weight-gradient launch (1x1 layers; conv_wgrad.hip: two gradient tensors)."""
    B, H, W, Cin = x.shape
    c1, c2 = dy1.shape[-1], dy2.shape[-1]
    spec = ConvSpec(Cin, c1 + c2, 1, 1, 0, 1)
    d = conv_desc(x.dtype, B, H, W, spec, None)
    L = _lib.lib()
    ws = workspace(L.sl_conv2d_bwd_weight_workspace(C.byref(d)), x.device, 'wgrad')
    dw = torch.empty((c1 + c2, Cin), dtype=torch.float32, device=x.device)
    part = _f32((_bias_rows(d, 0, 0), c1 + c2), x.device)
    tok = PROFILER.begin('conv_wgrad', d)
    check(L.sl_conv2d_bwd_weight_dy2(C.byref(d), _p(x), _p(dy1), _p(dy2), c1, _p(dw), _p(ws), ws.numel(), _p(part), _s()), 'conv2d_bwd_weight_dy2')
    PROFILER.end(tok)
    return dw, colsum(part).contiguous()


def conv2d_bwd_data_bnstat_folded(g, x, wext, bias, spec, gate, bn_x, mean, invstd):
    """Data gradient of the 1x1 conv `spec` (input x, gated incoming gradient g of its BatchNorm's output) with that BatchNorm's backward apply pass folded into the weights
    (bn_fold_weights); the result is gated with `gate` and reduced against bn_x as in conv2d_bwd_data_bnstat.  -> (dx, partial)."""
    B, H, W, _ = x.shape
    d = conv_desc(x.dtype, B, H, W, spec, None)
    L = _lib.lib()
    rows = L.sl_conv2d_bwd_data_bnstat_folded_rows(C.byref(d))
    assert rows > 0
    dx = torch.empty((B, H, W, spec.cin), dtype=x.dtype, device=x.device)
    part = _f32((rows, 2, spec.cin), x.device)
    tok = PROFILER.begin('conv_dgrad', d, EPI_GATE)
    check(L.sl_conv2d_bwd_data_bnstat_folded(C.byref(d), _p(g), _p(x), _p(wext), _p(bias), _p(gate), _p(bn_x), _p(mean), _p(invstd), _p(dx), _p(part), _s()),
          'conv2d_bwd_data_bnstat_folded')
    if tok is not None:
        PROFILER.end(tok, bn_x.numel() * bn_x.element_size() + gate.numel() + x.numel() * x.element_size())
    return dx, part


def bn_bwd2(dy, mask, x1, mean1, invstd1, gamma1, x2, mean2, invstd2, gamma2, outs1=(None, None), outs2=(None, None), pre_partials=None):
    """Train-mode backward of TWO BatchNorms whose outputs were added before one ReLU (bn3 + downsample BN, resnet.py:71-76): both see the gradient
    dy gated by `mask`; dy and the bits are swept once per pass for both.  -> (dx1, dgamma1, dbeta1, dx2, dgamma2, dbeta2).
    pre_partials = (partial1, partial2): dy is ALREADY gated (mask must be None) and both column-sum partials were produced by the data-gradient epilogue that wrote it
    (conv2d_bwd_data_addend_bnstat2): no reduce pass runs here."""
    Cn = x1.shape[-1]
    rows = x1.numel() // Cn
    L = _lib.lib()
    nb = x1.numel() * x1.element_size()
    if pre_partials is not None:
        assert mask is None
        part, nblk = pre_partials, pre_partials[0].shape[0]
    else:
        nblk = L.sl_bn_bwd_reduce_rows(rows, Cn)
        part = _f32((2, nblk, 2, Cn), x1.device)
        tok = PROFILER.begin_bytes('bn_bwd_reduce', 3 * nb + mask.numel())
        check(L.sl_bn_bwd_reduce2(dt(x1), _p(dy), _p(mask), _p(x1), _p(mean1), _p(invstd1), _p(part[0]), _p(x2), _p(mean2), _p(invstd2), _p(part[1]), rows, Cn, _s()),
              'bn_bwd_reduce2')
        PROFILER.end_bytes(tok)
    o = _f32((2, 5, Cn), x1.device)
    res = []
    for k, (gamma, mean, invstd, outs) in enumerate(((gamma1, mean1, invstd1, outs1), (gamma2, mean2, invstd2, outs2))):
        dg = o[k, 0] if outs[0] is None else outs[0]
        db = o[k, 1] if outs[1] is None else outs[1]
        check(L.sl_bn_bwd_finalize(_p(part[k]), nblk, Cn, rows, _p(gamma), _p(mean), _p(invstd), 1, _p(dg), _p(db), _p(o[k, 2]), _p(o[k, 3]), _p(o[k, 4]), _s()),
              'bn_bwd_finalize')
        res.append((dg, db))
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    tok = PROFILER.begin_bytes('bn_bwd_apply', 5 * nb + (mask.numel() if mask is not None else 0))
    check(L.sl_bn_bwd_apply2(dt(x1), _p(dy), _p(mask), _p(x1), _p(o[0, 2]), _p(o[0, 3]), _p(o[0, 4]), _p(mean1), _p(dx1),
                             _p(x2), _p(o[1, 2]), _p(o[1, 3]), _p(o[1, 4]), _p(mean2), _p(dx2), rows, Cn, _s()), 'bn_bwd_apply2')
    PROFILER.end_bytes(tok)
    return dx1, res[0][0], res[0][1], dx2, res[1][0], res[1][1]


def colsum_rows(t, batch=None):
    """Per-channel sum over all rows of an NHWC tensor (nn.Linear / conv bias gradient): csrc/pop_head.hip colsum_rows_partial + a fixed-order finalize
    (batch: an ops.ColsumBatch whose run() finalizes several of them in one launch)."""
    Cn = t.shape[-1]
    rows = t.numel() // Cn
    L = _lib.lib()
    nblk = L.sl_colsum_rows_blocks(rows, Cn, dt(t))
    part = _f32((nblk, Cn), t.device)
    check(L.sl_colsum_rows_partial(dt(t), _p(t), rows, Cn, _p(part), _s()), 'colsum_rows_partial')
    if batch is not None:
        return batch.add(part)                        # filled by batch.run()
    return colsum(part).contiguous()


# --------------------------------------------------------------------------------------------- stem
def stem_conv_fwd(img, w, dtype, want_stats):
    B, _, H, W = img.shape
    L = _lib.lib()
    y = torch.empty((B, H // 2, W // 2, 64), dtype=dtype, device=img.device)
    part = _f32((L.sl_stem_conv_stat_rows(B, H, W), 2, 64), img.device) if want_stats else None
    need = L.sl_stem_conv_fwd_workspace(_DT[dtype])
    ws = torch.empty(need, dtype=torch.uint8, device=img.device) if need else None
    tok = PROFILER.begin_bytes('stem_conv_fwd', img.numel() * img.element_size() + y.numel() * y.element_size())
    check(L.sl_stem_conv_fwd(_DT[dtype], _p(img), _p(w), _p(y), _p(part), B, H, W, _p(ws), _s()), 'stem_conv_fwd')
    PROFILER.end_bytes(tok)
    return y, part


def stem_bn_relu_pool(c0, scale, shift, want_idx):
    B, Hc, Wc, _ = c0.shape
    pooled = torch.empty((B, Hc // 2, Wc // 2, 64), dtype=c0.dtype, device=c0.device)
    idx = torch.empty((B, Hc // 2, Wc // 2, 64), dtype=torch.uint8, device=c0.device) if want_idx else None
    tok = PROFILER.begin_bytes('stem_bn_relu_pool_fwd', c0.numel() * c0.element_size() + pooled.numel() * pooled.element_size() + (idx.numel() if idx is not None else 0))
    check(_lib.lib().sl_stem_bn_relu_pool_fwd(dt(c0), _p(c0), _p(scale), _p(shift), _p(pooled), _p(idx), B, Hc, Wc, _s()), 'stem_bn_relu_pool_fwd')
    PROFILER.end_bytes(tok)
    return pooled, idx


def stem_pool_relu_bwd(dpooled, idx, c0, scale, shift):
    B, Hc, Wc, _ = c0.shape
    g0 = torch.empty_like(c0)
    check(_lib.lib().sl_stem_pool_relu_bwd(dt(c0), _p(dpooled), _p(idx), _p(c0), _p(scale), _p(shift), _p(g0), B, Hc, Wc, _s()), 'stem_pool_relu_bwd')
    return g0


def stem_pool_relu_bwd_bnstat(dpooled, idx, c0, scale, shift, mean, invstd):
    """stem_pool_relu_bwd + bn1's backward column sums in the same sweep -> (g0, partial [rows][2][64]) (bn_bwd(pre_partial=...))."""
    B, Hc, Wc, _ = c0.shape
    L = _lib.lib()
    g0 = torch.empty_like(c0)
    part = _f32((L.sl_stem_pool_relu_bwd_bnstat_rows(B, Hc, Wc), 2, 64), c0.device)
    nb = c0.numel() * c0.element_size()
    tok = PROFILER.begin_bytes('stem_pool_relu_bwd', 2 * nb + nb // 4 + dpooled.numel())
    check(L.sl_stem_pool_relu_bwd_bnstat(dt(c0), _p(dpooled), _p(idx), _p(c0), _p(scale), _p(shift), _p(mean), _p(invstd), _p(g0), _p(part), B, Hc, Wc, _s()),
          'stem_pool_relu_bwd_bnstat')
    PROFILER.end_bytes(tok)
    return g0, part


def stem_conv_bwd_weight(img, dc0):
    """7x7 stem weight gradient.  bf16: one MFMA kernel that gathers the im2col view from the image patch in the LDS (+ the fixed-order block
    reduce); fp32: im2col + the exact-fp32 MFMA wgrad kernel."""
    if dc0.dtype == torch.bfloat16:
        return stem_conv_bwd_weight_direct(img, dc0)
    return stem_conv_bwd_weight_im2col(img, dc0)


def stem_conv_bwd_weight_im2col(img, dc0):
    """im2col (materialised, [M,192]) + the conv wgrad kernel."""
    B, _, H, W = img.shape
    M = B * (H // 2) * (W // 2)
    col = torch.empty((1, 1, M, 192), dtype=dc0.dtype, device=img.device)
    check(_lib.lib().sl_stem_im2col(dt(dc0), _p(img), _p(col), B, H, W, _s()), 'stem_im2col')
    dw = conv2d_bwd_weight(col, dc0.view(1, 1, M, 64), ConvSpec(192, 64, 1))
    return dw.view(64, 192)[:, :147].reshape(64, 3, 7, 7).contiguous()


def stem_conv_bwd_weight_direct(img, dc0):
    B, _, H, W = img.shape
    L = _lib.lib()
    ws = workspace(L.sl_stem_conv_bwd_weight_workspace(B, H, W), img.device)
    dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=img.device)
    tok = PROFILER.begin_bytes('stem_conv_bwd_weight', img.numel() * img.element_size() + dc0.numel() * dc0.element_size())
    check(L.sl_stem_conv_bwd_weight(dt(dc0), _p(img), _p(dc0), _p(dw), _p(ws), ws.numel(), B, H, W, _s()), 'stem_conv_bwd_weight')
    PROFILER.end_bytes(tok)
    return dw


# --------------------------------------------------------------------------------------------- pyramid pooling
def ppm_desc(x, sizes):
    B, H, W, Cn = x.shape
    return SlPpmDesc(dt(x), B, H, W, Cn, len(sizes), (C.c_int * 4)(*(list(sizes) + [0] * (4 - len(sizes)))))


def ppm_rows(B, sizes):
    return B * sum(s * s for s in sizes)


def ppm_pool_fwd(x, sizes):
    d = ppm_desc(x, sizes)
    L = _lib.lib()
    ws = workspace(L.sl_ppm_workspace(C.byref(d)), x.device)
    pooled = torch.empty((ppm_rows(x.shape[0], sizes), x.shape[3]), dtype=torch.float32, device=x.device)
    tok = PROFILER.begin_bytes('ppm_pool_fwd', x.numel() * x.element_size())
    check(L.sl_ppm_pool_fwd(C.byref(d), _p(x), _p(pooled), _p(ws), ws.numel(), _s()), 'ppm_pool_fwd')
    PROFILER.end_bytes(tok)
    return pooled


def ppm_pool_bwd(dpooled, x_shape, dtype, sizes, dcat=None, cat_off=0):
    B, H, W, Cn = x_shape
    d = SlPpmDesc(_DT[dtype], B, H, W, Cn, len(sizes), (C.c_int * 4)(*(list(sizes) + [0] * (4 - len(sizes)))))
    assert dpooled.dtype == torch.float32
    dx = torch.empty((B, H, W, Cn), dtype=dtype, device=dpooled.device)
    pitch = dcat.shape[-1] if dcat is not None else 0
    L = _lib.lib()
    ws = workspace(L.sl_ppm_workspace(C.byref(d)), dpooled.device)
    tok = PROFILER.begin_bytes('ppm_pool_bwd', dx.numel() * dx.element_size() * (2 if dcat is not None else 1))
    check(L.sl_ppm_pool_bwd(C.byref(d), _p(dpooled), _p(dcat), pitch, cat_off, _p(dx), _p(ws), ws.numel(), _s()), 'ppm_pool_bwd')
    PROFILER.end_bytes(tok)
    return dx


def ppm_stage_bn_bwd(dy, y, x, B, sizes, means, invstds, gammas, trains, dgammas, dbetas, out=None):
    """BatchNorm + ReLU backward of all pyramid stages in one launch (csrc/ppm.hip ppm_stage_bn_bwd_kernel).  dy / y / x: float [rows][C]; per level lists of
    mean / invstd / gamma / train flag and of the destinations dgamma / dbeta (float [C] tensors).  -> dx [rows][C]."""
    n = len(sizes)
    Cn = x.shape[1]
    assert dy.dtype == y.dtype == x.dtype == torch.float32 and dy.is_contiguous() and y.is_contiguous() and x.is_contiguous() and x.shape[0] == ppm_rows(B, sizes)
    d = SlPpmDesc(SL_F32, B, 0, 0, Cn, n, (C.c_int * 4)(*(list(sizes) + [0] * (4 - n))))
    arr = lambda ts: (C.c_void_p * n)(*[_p(t) for t in ts])
    dx = out if out is not None else torch.empty_like(x)
    check(_lib.lib().sl_ppm_stage_bn_bwd(C.byref(d), Cn, _p(dy), _p(y), _p(x), arr(means), arr(invstds), arr(gammas), (C.c_int * n)(*[int(t) for t in trains]),
                                         arr(dgammas), arr(dbetas), _p(dx), _s()), 'ppm_stage_bn_bwd')
    return dx


def ppm_upsample_fwd(stage, x_shape, sizes, dtype):
    B, H, W, Cn = x_shape
    Cs = stage.shape[1]
    assert stage.dtype == torch.float32
    d = SlPpmDesc(_DT[dtype], B, H, W, Cn, len(sizes), (C.c_int * 4)(*(list(sizes) + [0] * (4 - len(sizes)))))
    priors = torch.empty((B, H, W, len(sizes) * Cs), dtype=dtype, device=stage.device)
    check(_lib.lib().sl_ppm_upsample_fwd(C.byref(d), Cs, _p(stage), _p(priors), _s()), 'ppm_upsample_fwd')
    return priors


def ppm_upsample_bwd(dcat, x_shape, sizes, Cs):
    B, H, W, Cn = x_shape
    L = _lib.lib()
    d = SlPpmDesc(dt(dcat), B, H, W, Cn, len(sizes), (C.c_int * 4)(*(list(sizes) + [0] * (4 - len(sizes)))))
    ws = workspace(L.sl_ppm_workspace(C.byref(d)), dcat.device)
    dstage = torch.empty((ppm_rows(B, sizes), Cs), dtype=torch.float32, device=dcat.device)
    check(L.sl_ppm_upsample_bwd(C.byref(d), Cs, _p(dcat), dcat.shape[-1], _p(dstage), _p(ws), ws.numel(), _s()), 'ppm_upsample_bwd')
    return dstage


def weight_prep_slice(w, dtype, ci_off, ci_cnt):
    O, I, KH, KW = w.shape
    wf = torch.empty((O, KH, KW, ci_cnt), dtype=dtype, device=w.device)
    wb = torch.empty((ci_cnt, KH, KW, O), dtype=dtype, device=w.device)
    check(_lib.lib().sl_weight_prep_slice(_DT[dtype], _p(w.detach()), O, I, ci_off, ci_cnt, KH, KW, _p(wf), _p(wb), _s()), 'weight_prep_slice')
    return wf, wb


def ppm_wq_prep(w, Cs, nl):
    N, Ctot = w.shape[0], w.shape[1]
    wq_f = _f32((nl, 9 * N, Cs), w.device)
    wq_b = _f32((nl, Cs, 9 * N), w.device)
    check(_lib.lib().sl_ppm_wq_prep(_p(w.detach()), N, Ctot, Cs, nl, _p(wq_f), _p(wq_b), _s()), 'ppm_wq_prep')
    return wq_f, wq_b


def ppm_dwq_scatter(dwq, dw_full, Cs, nl):
    N, Ctot = dw_full.shape[0], dw_full.shape[1]
    check(_lib.lib().sl_ppm_dwq_scatter(_p(dwq), N, Ctot, Cs, nl, _p(dw_full), _s()), 'ppm_dwq_scatter')


def _ppm_d(dtype, x_shape, sizes):
    B, H, W, Cn = x_shape
    return SlPpmDesc(_DT[dtype], B, H, W, Cn, len(sizes), (C.c_int * 4)(*(list(sizes) + [0] * (4 - len(sizes)))))


def ppm_rows_gemm(x, w, B, sizes, want_stats=False):
    """y[r] = w[level(r)] @ x[r] over the pyramid rows; x [rows][K] float, w [nl][N][K] float -- or a list of nl float tensors of N * K elements each (the levels' own
    weight copies, row-major [N][K]) with N given by the first one's leading dimension.  Returns (y, stat partials or None)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    rows, K = x.shape
    if isinstance(w, (list, tuple)):
        nl, N = len(w), w[0].numel() // K
        for t in w:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == N * K
    else:
        assert w.dtype == torch.float32 and w.is_contiguous()
        nl, N, K2 = w.shape
        assert K2 == K
        w = [w[l] for l in range(nl)]
    assert nl == len(sizes) and rows == ppm_rows(B, sizes)
    d = SlPpmDesc(SL_F32, B, 1, 1, 8, nl, (C.c_int * 4)(*(list(sizes) + [0] * (4 - nl))))
    L = _lib.lib()
    ws = workspace(L.sl_ppm_rows_gemm_workspace(C.byref(d), K, N), x.device)
    y = _f32((rows, N), x.device)
    part = _f32((L.sl_ppm_rows_gemm_stat_rows(C.byref(d)), 2, N), x.device) if want_stats else None
    check(L.sl_ppm_rows_gemm_levels(C.byref(d), K, N, _p(x), (C.c_void_p * nl)(*[_p(t) for t in w]), _p(y), _p(part), _p(ws), ws.numel(), _s()), 'ppm_rows_gemm')
    return y, part


def ppm_rows_wgrad(a, x, B, sizes, outs=None):
    """Weight gradients of ppm_rows_gemm for all levels in ONE launch: dw[l] [N][K] = sum over level l's rows of a[r][:]^T x[r][:].  a [rows][N] (gradient wrt the GEMM's
    output), x [rows][K] (its input), float.  outs: per level a float [N*K] destination (a parameter's gradient buffer) or None -> a fresh [nl][N][K] tensor's slices."""
    assert a.dtype == torch.float32 and x.dtype == torch.float32 and a.is_contiguous() and x.is_contiguous() and a.shape[0] == x.shape[0] == ppm_rows(B, sizes)
    nl, N, K = len(sizes), a.shape[1], x.shape[1]
    fresh = _f32((nl, N, K), a.device) if (outs is None or any(o is None for o in outs)) else None
    dws = [fresh[l] if (outs is None or outs[l] is None) else outs[l] for l in range(nl)]
    for t in dws:
        assert t.numel() == N * K and t.dtype == torch.float32 and t.is_contiguous()
    d = SlPpmDesc(SL_F32, B, 1, 1, 8, nl, (C.c_int * 4)(*(list(sizes) + [0] * (4 - nl))))
    tok = PROFILER.begin_bytes('ppm_rows_wgrad', (a.numel() + x.numel() + nl * N * K) * 4)
    check(_lib.lib().sl_ppm_rows_wgrad(C.byref(d), N, K, _p(a), _p(x), (C.c_void_p * nl)(*[_p(t) for t in dws]), _s()), 'ppm_rows_wgrad')
    PROFILER.end_bytes(tok)
    return dws


def ppm_stat_groups(B, sizes):
    """Prefix of the 128-row statistic groups per level in ppm_rows_gemm's partials."""
    off = [0]
    for s in sizes:
        off.append(off[-1] + (B * s * s + 127) // 128)
    return off


def ppm_fact_gather(q, x_shape, sizes, N, dtype):
    d = _ppm_d(dtype, x_shape, sizes)
    L = _lib.lib()
    ws = workspace(L.sl_ppm_fact_workspace(C.byref(d), N), q.device)
    B, H, W, _ = x_shape
    g = torch.empty((B, H, W, N), dtype=dtype, device=q.device)
    tok = PROFILER.begin_bytes('ppm_fact_gather', g.numel() * g.element_size() + q.numel() * 4)
    check(L.sl_ppm_fact_gather(C.byref(d), N, _p(q), _p(g), _p(ws), ws.numel(), _s()), 'ppm_fact_gather')
    PROFILER.end_bytes(tok)
    return g


def ppm_fact_scatter(dcb, x_shape, sizes):
    N = dcb.shape[-1]
    d = _ppm_d(dcb.dtype, x_shape, sizes)
    L = _lib.lib()
    ws = workspace(L.sl_ppm_fact_workspace(C.byref(d), N), dcb.device)
    gq = _f32((ppm_rows(x_shape[0], sizes), 9 * N), dcb.device)
    tok = PROFILER.begin_bytes('ppm_fact_scatter', dcb.numel() * dcb.element_size() + gq.numel() * 4)
    check(L.sl_ppm_fact_scatter(C.byref(d), N, _p(dcb), _p(gq), _p(ws), ws.numel(), _s()), 'ppm_fact_scatter')
    PROFILER.end_bytes(tok)
    return gq


# --------------------------------------------------------------------------------------------- POP head
def pop_decompose_into(feats2d, S, bg_out):
    R, Cn = feats2d.shape
    Kt = S.shape[0]
    proj = _f32((R, Kt), feats2d.device)
    tok = PROFILER.begin_bytes('pop_decompose_fwd', 2 * feats2d.numel() * feats2d.element_size() + proj.numel() * 4)
    check(_lib.lib().sl_pop_decompose_fwd(dt(feats2d), _p(feats2d), _p(S), Kt, _p(proj), _p(bg_out), R, Cn, _s()), 'pop_decompose_fwd')
    PROFILER.end_bytes(tok)
    return proj


def pop_proto_rows(S, dst):
    Kt, Cn = S.shape
    check(_lib.lib().sl_pop_proto_rows(dt(dst), _p(S), Kt, Cn, _p(dst), _s()), 'pop_proto_rows')


def rowdot_fwd(h, w):
    R, Cn = h.shape
    z = _f32((R,), h.device)
    tok = PROFILER.begin_bytes('rowdot_fwd', h.numel() * h.element_size())
    check(_lib.lib().sl_rowdot_fwd(dt(h), _p(h), _p(w), _p(z), R, Cn, _s()), 'rowdot_fwd')
    PROFILER.end_bytes(tok)
    return z


def colsum(part):
    nblk, Cn = part.shape[0], part[0].numel()
    out = _f32(tuple(part.shape[1:]), part.device)
    check(_lib.lib().sl_colsum_finalize(_p(part), nblk, Cn, _p(out), _s()), 'colsum_finalize')
    return out


class ColsumBatch:
    """Column sums of several partial buffers with ONE finalize launch: add(part [nblk, ...]) hands out the result tensor right away (shape
    part.shape[1:]), run() fills all of them (sl_colsum_finalize_multi; SL_COLSUM_MAX entries per launch)."""

    def __init__(self):
        self.items = []

    def add(self, part):
        out = _f32(tuple(part.shape[1:]), part.device)
        self.items.append((part, out))
        return out

    def run(self):
        L = _lib.lib()
        for i0 in range(0, len(self.items), _lib.SL_COLSUM_MAX):
            chunk = self.items[i0:i0 + _lib.SL_COLSUM_MAX]
            b = _lib.SlColsumBatch()
            b.n = len(chunk)
            for i, (part, out) in enumerate(chunk):
                b.part[i], b.out[i], b.nblk[i], b.C[i] = part.data_ptr(), out.data_ptr(), part.shape[0], part[0].numel()
            check(L.sl_colsum_finalize_multi(C.byref(b), _s()), 'colsum_finalize_multi')
        self.items = []


def allreduce_partials(part):
    """SyncBatchNorm: [nblk][...] float statistic partials -> [2][...] float rows (hi, lo) of the GLOBAL totals: column sums in double,
    ONE all-reduce (double) over the process group, split back for the finalize kernels (which add partial rows in double)."""
    import torch.distributed as dist
    nblk, n = part.shape[0], part[0].numel()
    tot = torch.empty(tuple(part.shape[1:]), dtype=torch.float64, device=part.device)
    check(_lib.lib().sl_colsum_f64(_p(part), nblk, n, _p(tot), _s()), 'colsum_f64')
    dist.all_reduce(tot)
    out = _f32((2,) + tuple(part.shape[1:]), part.device)
    check(_lib.lib().sl_f64_split(_p(tot), n, _p(out), _s()), 'f64_split')
    return out


def rowdot_bwd(h, w, dz):
    R, Cn = h.shape
    L = _lib.lib()
    nblk = L.sl_rowdot_bwd_rows(R, Cn)
    part = _f32((nblk, Cn), h.device)
    dh = torch.empty_like(h)
    tok = PROFILER.begin_bytes('rowdot_bwd', 2 * h.numel() * h.element_size())
    check(L.sl_rowdot_bwd(dt(h), _p(h), _p(w), _p(dz), _p(dh), _p(part), R, Cn, _s()), 'rowdot_bwd')
    PROFILER.end_bytes(tok)
    return dh, colsum(part)


def pop_combine_fwd(proj, z_bg, a, b, B, N):
    Kt = proj.shape[1]
    preds = _f32((B, 1 + Kt, N), proj.device)
    check(_lib.lib().sl_pop_combine_fwd(_p(proj), _p(z_bg), _p(a), _p(b), Kt, _p(preds), B, N, _s()), 'pop_combine_fwd')
    return preds


def pop_combine_bwd(dpreds, proj, a, b, B, N):
    Kt = proj.shape[1]
    L = _lib.lib()
    nblk = L.sl_pop_combine_bwd_rows(B, N)
    dz = _f32((B * N,), proj.device)
    dproj = torch.empty_like(proj)
    part = _f32((nblk, 2 * Kt), proj.device)
    check(L.sl_pop_combine_bwd(_p(dpreds), _p(proj), _p(a), _p(b), Kt, _p(dz), _p(dproj), _p(part), B, N, _s()), 'pop_combine_bwd')
    dab = colsum(part)
    return dz, dproj, dab[:Kt], dab[Kt:]


def pop_decompose_bwd(dg, feats2d, S, proj, dproj):
    R, Cn = feats2d.shape
    Kt = S.shape[0]
    L = _lib.lib()
    nblk = L.sl_pop_decompose_bwd_rows(R)
    part = _f32((nblk, Kt, Cn), feats2d.device)
    dq = torch.empty_like(feats2d)
    tok = PROFILER.begin_bytes('pop_decompose_bwd', 3 * feats2d.numel() * feats2d.element_size() + 2 * proj.numel() * 4)
    check(L.sl_pop_decompose_bwd(dt(feats2d), _p(dg), _p(feats2d), _p(S), _p(proj), _p(dproj), Kt, _p(dq), _p(part), R, Cn, _s()), 'pop_decompose_bwd')
    PROFILER.end_bytes(tok)
    return dq, colsum(part)


# --------------------------------------------------------------------------------------------- loss & labels
def upsample_ce_fwd(logits, target, ignore_index):
    B, K, h, w = logits.shape
    H, W = target.shape[1:]
    L = _lib.lib()
    nblk = L.sl_upsample_ce_rows(B, H, W)
    part = _f32((nblk, 2), logits.device)
    tok = PROFILER.begin_bytes('upsample_ce_fwd', logits.numel() * 4 + target.numel() * target.element_size())
    check(L.sl_upsample_ce_fwd(_p(logits), _p(target), B, K, h, w, H, W, ignore_index, _p(part), _s()), 'upsample_ce_fwd')
    PROFILER.end_bytes(tok)
    out = _f32((2,), logits.device)
    check(L.sl_upsample_ce_finalize(_p(part), nblk, _p(out), _s()), 'upsample_ce_finalize')
    return out


def upsample_ce_bwd(logits, target, loss_cnt, gscale, ignore_index):
    B, K, h, w = logits.shape
    H, W = target.shape[1:]
    dl = torch.empty_like(logits)
    tok = PROFILER.begin_bytes('upsample_ce_bwd', 2 * logits.numel() * 4 + target.numel() * target.element_size())
    check(_lib.lib().sl_upsample_ce_bwd(_p(logits), _p(target), _p(loss_cnt), _p(gscale), B, K, h, w, H, W, ignore_index, _p(dl), _s()), 'upsample_ce_bwd')
    PROFILER.end_bytes(tok)
    return dl


def pseudo_label_(logits, mask, n_base):
    """In place on `mask` (int64 [B,H,W]); logits [B,K2,h,w] float."""
    B, K2, h, w = logits.shape
    check(_lib.lib().sl_pseudo_label(_p(logits), K2, h, w, _p(mask), B, mask.shape[1], mask.shape[2], n_base, _s()), 'pseudo_label')
    return mask


def upsample_argmax(logits, size):
    B, K, h, w = logits.shape
    out = torch.empty((B, size[0], size[1]), dtype=torch.uint8, device=logits.device)
    check(_lib.lib().sl_upsample_argmax(_p(logits), B, K, h, w, size[0], size[1], _p(out), _s()), 'upsample_argmax')
    return out


def upsample_logits(logits, size):
    """F.interpolate(logits, size, mode='bilinear', align_corners=True) -- the array eval_base.py:189-190 dumps per tile."""
    B, K, h, w = logits.shape
    out = _f32((B, K, size[0], size[1]), logits.device)
    check(_lib.lib().sl_upsample_logits(_p(logits), B, K, h, w, size[0], size[1], _p(out), _s()), 'upsample_logits')
    return out


def fuse_argmax(mats):
    """argmax over classes of the mean of probability maps (list of [K,H,W] float GPU tensors), fusemat.py:35-52 -> uint8 [H,W]."""
    K, H, W = mats[0].shape
    for m in mats:
        if tuple(m.shape) != (K, H, W) or m.dtype != torch.float32:
            raise RuntimeError('fuse_argmax: every probability map must be float32 [%d,%d,%d]' % (K, H, W))
    import struct
    table = torch.frombuffer(bytearray(struct.pack('<%dQ' % len(mats), *[m.data_ptr() for m in mats])), dtype=torch.uint8).to(mats[0].device)
    out = torch.empty((H, W), dtype=torch.uint8, device=mats[0].device)
    for m in mats:
        _p(m)
    check(_lib.lib().sl_fuse_argmax(_p(table), len(mats), K, H * W, _p(out), _s()), 'fuse_argmax')
    return out


def iou_hist(pred_u8, target, K, ignore_index):
    hist = torch.zeros((3, K), dtype=torch.int64, device=target.device)
    check(_lib.lib().sl_iou_hist(_p(pred_u8), _p(target), target.numel(), K, ignore_index, _p(hist), _s()), 'iou_hist')
    return hist


def adamw_multi(table, n, total_chunks, b1, b2, eps, bc1, rs2, bc1n, rs2n, repeat, grad_scale=None):
    check(_lib.lib().sl_adamw_multi(_p(table), n, int(total_chunks), b1, b2, eps, bc1, rs2, bc1n, rs2n, int(repeat), _p(grad_scale), _s()), 'adamw_multi')


def adamw_multi_dev(table, n, total_chunks, b1, b2, eps, hyper, repeat, grad_scale=None):
    check(_lib.lib().sl_adamw_multi_dev(_p(table), n, int(total_chunks), b1, b2, eps, _p(hyper), int(repeat), _p(grad_scale), _s()), 'adamw_multi_dev')


def sgd_multi(table, n, total_chunks, momentum, hyper=None, grad_scale=None):
    check(_lib.lib().sl_sgd_multi(_p(table), n, int(total_chunks), float(momentum), _p(hyper), _p(grad_scale), _s()), 'sgd_multi')


def grad_norm_coef(grads, max_norm, grad_div=1):
    """(norm, coef) of clip_grad_norm_(max_norm) over a list of contiguous float32 GPU gradient tensors: ceil(n / 64) norm launches + one finalize (csrc/optim.hip).
    Both results are views of one 2-element float tensor.  grad_div > 1: the gradients hold the SUM over that many ranks (see optim.clip_coefficient)."""
    L = _lib.lib()
    dev = grads[0].device
    chunks = [(g.numel() + 4095) // 4096 for g in grads]
    total = sum(chunks)
    partial = _f32((total,), dev)
    base = 0
    for i0 in range(0, len(grads), _lib.SL_NORM_MAX):
        part = grads[i0:i0 + _lib.SL_NORM_MAX]
        b = _lib.SlNormBatch()
        b.n, b.chunk_base, c = len(part), base, 0
        for i, g in enumerate(part):
            b.grad[i], b.numel[i], b.chunk0[i] = _p(g), g.numel(), c
            c += chunks[i0 + i]
        b.chunk0[len(part)] = c
        check(L.sl_grad_sqnorm_multi(C.byref(b), _p(partial), _s()), 'grad_sqnorm_multi')
        base += c
    out = _f32((2,), dev)
    check(L.sl_grad_norm_finalize(_p(partial), total, float(max_norm), 1.0 / float(grad_div), _p(out), _s()), 'grad_norm_finalize')
    return out[0], out[1:2]


def store_floats(dst, values):
    """values (<= 16 python floats) -> the first len(values) elements of the float32 GPU tensor dst, as kernel arguments (no host -> device copy)."""
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    check(_lib.lib().sl_store_floats(_p(dst), len(values), arr, _s()), 'store_floats')


def confusion_matrix(pred_u8, target, K, ignore_index):
    """[K][K] int64 counts, rows = ground truth, columns = prediction, pixels with target == ignore_index dropped."""
    cm = torch.zeros((K, K), dtype=torch.int64, device=target.device)
    check(_lib.lib().sl_confusion_matrix(_p(pred_u8), _p(target), target.numel(), K, ignore_index, _p(cm), _s()), 'confusion_matrix')
    return cm


def masked_avg_pool(feature_nhwc, mask):
    B, h, w, Cn = feature_nhwc.shape
    H, W = mask.shape[-2:]
    proto = _f32((Cn,), mask.device)
    check(_lib.lib().sl_masked_avg_pool(dt(feature_nhwc), _p(feature_nhwc), _p(mask), B, h, w, Cn, H, W, _p(proto), _s()), 'masked_avg_pool')
    return proto


def nhwc_to_nchw_f32(x):
    B, H, W, Cn = x.shape
    out = _f32((B, Cn, H, W), x.device)
    check(_lib.lib().sl_nhwc_to_nchw_f32(dt(x), _p(x), _p(out), B, H, W, Cn, _s()), 'nhwc_to_nchw_f32')
    return out


def nchw_f32_to_nhwc(x, dtype):
    B, Cn, H, W = x.shape
    out = torch.empty((B, H, W, Cn), dtype=dtype, device=x.device)
    check(_lib.lib().sl_nchw_f32_to_nhwc(_DT[dtype], _p(x), _p(out), B, H, W, Cn, _s()), 'nchw_f32_to_nhwc')
    return out


# --------------------------------------------------------------------------------------------- prototype preparation
def proto_fused_ok(Ka, Kb, Cn):
    """Do the one-block prototype kernels (forward AND backward) hold Ka + Kb prototypes of Cn channels?"""
    return bool(_lib.lib().sl_pop_proto_ok(int(Ka), int(Kb), int(Cn)))


def pop_proto_fwd(Ea, Eb=None):
    """(Sa, Sb, inv_norm, G, orth): L2-normalised prototypes, G = Sa [Sa ; Sb]^T and mean |G[i][j]|, j > i, in one launch."""
    Ka, Cn = Ea.shape
    Kb = 0 if Eb is None else Eb.shape[0]
    Sa, Sb = torch.empty_like(Ea), (None if Eb is None else torch.empty_like(Eb))
    aux = _f32((Ka + Kb + Ka * (Ka + Kb) + 1,), Ea.device)
    inv, G, orth = aux[:Ka + Kb], aux[Ka + Kb:Ka + Kb + Ka * (Ka + Kb)], aux[Ka + Kb + Ka * (Ka + Kb):]
    check(_lib.lib().sl_pop_proto_fwd(_p(Ea), Ka, _p(Eb), Kb, Cn, _p(Sa), _p(Sb), _p(inv), _p(G), _p(orth), _s()), 'pop_proto_fwd')
    return Sa, Sb, inv, G, orth


def pop_proto_bwd(Sa, Sb, inv, G, dSa, dSb, dorth, need_a=True, need_b=False):
    Ka, Cn = Sa.shape
    Kb = 0 if Sb is None else Sb.shape[0]
    dEa = torch.empty_like(Sa) if need_a else None
    dEb = torch.empty_like(Sb) if (need_b and Sb is not None) else None
    check(_lib.lib().sl_pop_proto_bwd(_p(Sa), Ka, _p(Sb), Kb, Cn, _p(inv), _p(G), _p(dSa), _p(dSb), _p(dorth), _p(dEa), _p(dEb), _s()), 'pop_proto_bwd')
    return dEa, dEb


def copy2d_multi(table, n, total_chunks):
    check(_lib.lib().sl_copy2d_multi(_p(table), n, int(total_chunks), _s()), 'copy2d_multi')


def relpos_gather_multi(table, n):
    check(_lib.lib().sl_relpos_gather_multi(_p(table), n, _s()), 'relpos_gather_multi')
