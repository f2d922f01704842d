"""Round-4 gates: the advisor's findings of round 3 (two live forward passes through the cross-block BatchNorm fusion, the shape check of the fused prototype
kernels), small-M conv dispatch (the fine-tune pair's 8 192-row layers on split-K), the faster BatchNorm finalize kernels, the TIFF tile decode."""
import os

import numpy as np
import pytest
import torch

from oracle import formula as fm

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------------------------- advisor (round 3, medium): two forwards before the first backward
def test_two_live_forward_passes_keep_their_own_bn3_handover(hip):
    """functional.BottleneckFn hands ReLU bits / c3 of block i to block i + 1's backward and the column sums back (resnet.py:71-78 across a block boundary).  Round 3 kept
    that in module state read at BACKWARD time: loss1 = model(x1); loss2 = model(x2); loss1.backward() gated pass 1's gradient with pass 2's bits, silently.  The records
    are per forward pass now (functional._BlockLink in ctx): gradients of two interleaved passes equal the gradients of each pass run alone, bit for bit."""
    from segland_amd import ops
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    torch.manual_seed(5)
    m = GFSS_Model(n_base=7, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.bfloat16).to(DEV).train()
    for mod in m.modules():                              # running statistics do not enter a train-mode gradient, but keep the passes independent of their order anyway
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.0
    xs = [fm.formula_image(16, 256, 256, 'live/img%d' % i).to(DEV) for i in range(2)]
    ys = [fm.formula_mask(16, 256, 256, 8, 'live/mask%d' % i, block=16, ignore_rows=8 + 8 * i).to(DEV) for i in range(2)]
    calls = [0]
    real = ops.conv2d_bwd_data_addend_bnstat

    def counted(*a, **k):
        out = real(*a, **k)
        calls[0] += out is not None
        return out

    def grads():
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    try:
        ops.conv2d_bwd_data_addend_bnstat = counted
        alone = []
        for i in range(2):
            m.zero_grad(set_to_none=True)
            m(xs[i], ys[i])['total_loss'].backward()
            alone.append(grads())
        assert calls[0] >= 2, 'the cross-block route was not taken at this shape (%d calls)' % calls[0]
        for order in ((0, 1), (1, 0)):
            losses = [m(xs[i], ys[i])['total_loss'] for i in range(2)]          # both graphs alive
            for i in order:
                m.zero_grad(set_to_none=True)
                losses[i].backward()
                g = grads()
                bad = [k for k in alone[i] if not torch.equal(g[k], alone[i][k])]
                assert not bad, 'pass %d (backward order %s): %d gradients differ from the pass run alone, e.g. %s' % (i, order, len(bad), bad[:3])
    finally:
        ops.conv2d_bwd_data_addend_bnstat = real


def test_prototype_kernels_refuse_what_their_backward_cannot_hold(hip):
    """Advisor (round 3, low): sl_pop_proto_fwd accepted (Ka + Kb) * C * 4 bytes up to 60 KB while the backward needs twice that: C = 512 with 16..29 prototypes passed the
    forward and raised mid-step.  sl_pop_proto_ok carries the backward's requirement; GFSS_Model._protos falls back to the torch ops, so --base-classes 15 trains."""
    from segland_amd import ops
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    assert ops.proto_fused_ok(7, 0, 512) and ops.proto_fused_ok(4, 7, 512) and ops.proto_fused_ok(14, 0, 512)
    assert not ops.proto_fused_ok(15, 0, 512) and not ops.proto_fused_ok(20, 9, 512) and ops.proto_fused_ok(20, 9, 128)
    E = torch.randn(15, 512, device=DEV)
    with pytest.raises(RuntimeError):
        ops.pop_proto_fwd(E)
    torch.manual_seed(2)
    m = GFSS_Model(n_base=15, criterion=OrthLoss(255), backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=torch.float32).to(DEV).train()
    img = fm.formula_image(2, 64, 64, 'p15/img').to(DEV)
    mask = fm.formula_mask(2, 64, 64, 16, 'p15/mask', block=8, ignore_rows=4).to(DEV)
    d = m(img, mask)
    d['total_loss'].backward()
    assert torch.isfinite(d['total_loss']) and m.base_emb.grad is not None and torch.isfinite(m.base_emb.grad).all() and float(d['orth_loss']) >= 0.0


# --------------------------------------------------------------------------------------------- config 4: inference convs on 8 192 pixel rows
SPLIT_CASES = [
    # Cin, Cout, k, dil, pre_addend, residual      (B = 2, 64 x 64: 8 192 rows.  Only long 3x3 layers are split -- the pyramid conv of pspnet_pop.py:23-29 on a pair)
    (2048, 512, 3, 1, True, False),
    (1024, 256, 3, 2, False, True),
    (2048, 256, 3, 4, False, False),
]


@pytest.mark.parametrize('cin,cout,k,dil,pre,res', SPLIT_CASES)
def test_split_k_inference_conv(hip, cin, cout, k, dil, pre, res):
    """Frozen 3x3 conv + folded BatchNorm (+ residual) + ReLU on 8 192 rows: 32-64 tiles of 16 x 16 pixels x 256 channels cannot fill 256 CUs, so a layer with >= 1024
    input channels is cut along K by 64-channel chunks (patch kernel), fp32 partial tiles go through a workspace and one finishing launch sums them in a fixed order and
    applies the epilogue (ft_pop.py:233-269: the frozen pyramid conv of a fine-tune pair, 251 -> 152 us).  Against fp32 torch on the bf16-rounded operands, against the
    unsplit dispatch (no workspace), and run to run bit for bit; shorter 3x3 layers and every 1x1 layer stay unsplit (measured slower split: tools/ft_shapes.py)."""
    import ctypes as C
    import torch.nn.functional as F
    from segland_amd import _lib, ops
    B, H, W = 2, 64, 64
    g = torch.Generator(device='cpu').manual_seed(cin + cout + k)
    dt = torch.bfloat16
    x = torch.randn(B, H, W, cin, generator=g).to(dt).to(DEV)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dt).float().to(DEV)
    scale = (torch.rand(cout, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(cout, generator=g) * 0.2).to(DEV)
    pa = torch.randn(B, H, W, cout, generator=g).to(dt).to(DEV) if pre else None
    rs = torch.randn(B, H, W, cout, generator=g).to(dt).to(DEV) if res else None
    spec = ops.ConvSpec(cin, cout, k, 1, dil * (k // 2), dil)
    wf, _ = ops.weight_prep(w, dt)
    L = _lib.lib()
    d = ops.conv_desc(dt, B, H, W, spec)
    need = L.sl_conv2d_affine_fwd_workspace(C.byref(d))
    assert need >= 2 * B * H * W * cout * 4, 'the layer is not split (%d bytes)' % need
    y = ops.conv2d_affine_fwd(x, wf, spec, scale, shift, residual=rs, relu=True, pre_addend=pa)
    y2 = ops.conv2d_affine_fwd(x, wf, spec, scale, shift, residual=rs, relu=True, pre_addend=pa)
    assert torch.equal(y, y2), 'split-K sums must be bit-stable'
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 1, dil * (k // 2), dil).permute(0, 2, 3, 1)
    ref = ref.to(dt).float()                                     # the kernels round the conv result before the epilogue
    if pre:
        ref = ref + pa.float()
    ref = ref * scale + shift
    if res:
        ref = ref + rs.float()
    ref = torch.relu(ref)
    err = float((y.float() - ref).abs().max())
    assert err <= 2e-2 * float(ref.abs().max()), (err, float(ref.abs().max()))
    # the unsplit dispatch of the same layer (no workspace handed in)
    y0 = torch.empty_like(y)
    _lib.check(L.sl_conv2d_affine_fwd_ex(C.byref(d), ops._p(x), None, ops._p(wf), ops._p(pa), ops._p(scale), ops._p(shift), ops._p(rs), 1, ops._p(y0), None, 0, ops._s()), 'unsplit')
    e0 = float((y.float() - y0.float()).abs().max())
    assert e0 <= 1.6e-2 * float(ref.abs().max()), e0            # two bf16 roundings of different K orders
    # not split: ragged row count, enough tiles, short K, 1x1
    assert L.sl_conv2d_affine_fwd_workspace(C.byref(ops.conv_desc(dt, 2, 60, 64, spec))) == 0
    assert L.sl_conv2d_affine_fwd_workspace(C.byref(ops.conv_desc(dt, 16, 64, 64, spec))) == 0
    assert L.sl_conv2d_affine_fwd_workspace(C.byref(ops.conv_desc(dt, 2, 64, 64, ops.ConvSpec(512, 512, 3, 1, 4, 4)))) == 0
    assert L.sl_conv2d_affine_fwd_workspace(C.byref(ops.conv_desc(dt, 2, 64, 64, ops.ConvSpec(2048, 512, 1, 1, 0, 1)))) == 0


def test_ft_pair_forward_bf16_matches_fp32_mode(hip):
    """Config 4 end to end: forward_all's frozen feature extractor on ONE tile pair (8 192 rows per layer from layer2 on) in bf16 -- the pyramid conv split along K, its prior
    term factorised as in training -- against the exact-fp32 mode of the same model (which takes none of the bf16-only routes)."""
    from segland_amd.loss.criterion import OrthLoss
    from segland_amd.networks.pspnet_pop import GFSS_Model
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(4)
        m = GFSS_Model(n_base=7, criterion=OrthLoss(255), is_ft=True, n_novel=4, backbone='resnet50', pretrained_model=None, dilated=True, os=8, compute_dtype=dt).to(DEV)
        m.init_cls_n()
        m.eval()
        with torch.no_grad():
            outs[dt] = m(torch.cat([fm.formula_image(1, 512, 512, 'ftp/a'), fm.formula_image(1, 512, 512, 'ftp/b')]).to(DEV)).float()
    a, b = outs[torch.bfloat16], outs[torch.float32]
    rel = float((a - b).norm() / b.norm())
    agree = float((a.argmax(1) == b.argmax(1)).float().mean())
    print('ft pair forward, bf16 vs fp32 mode: rel L2 %.4f, argmax agreement %.4f' % (rel, agree))
    assert rel <= 5e-2 and agree >= 0.97


# --------------------------------------------------------------------------------------------- f-2: what crosses the process boundary and the bus
def test_packed_row_cropped_tiles_prepare_identically(hip):
    """dataset/oem.py RawCollate / oem_ft.py PairCollate (they run in the DataLoader workers) cut a tile down to the rows its crop reads and pack the batch into ONE
    shared-memory buffer; TileAugmenter then makes one host -> device copy.  The prepared tensors equal those of the plain list-of-arrays path (the one goldens
    G17 / G19 pin bit for bit against dataset/base_dataset.py:29-175 of the reference) -- tiles larger, equal to and SMALLER than the crop, all flips / rotations."""
    from segland_amd.dataset.augment import TileAugmenter
    from segland_amd.dataset.oem import PackedTiles, RawCollate
    from segland_amd.dataset.oem_ft import PairAugmenter, PairCollate
    rng = np.random.RandomState(3)
    ch = cw = 64
    sizes = [(96, 80), (64, 64), (40, 100), (130, 64), (64, 50), (200, 72), (70, 70), (33, 21)]
    tiles = [(rng.randint(0, 256, (h, w, 3)).astype(np.uint8), rng.randint(0, 12, (h, w)).astype(np.uint8)) for h, w in sizes]
    prm = [(int(rng.randint(0, max(h - ch, 0) + 1)), int(rng.randint(0, max(w - cw, 0) + 1)), bool(i % 2), i % 4) for i, (h, w) in enumerate(sizes)]
    aug = TileAugmenter((ch, cw), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 255, device=DEV)
    want_i, want_l = aug.prepare(tiles, prm)
    batch = [(t[0], t[1], p, 'id%d' % i) for i, (t, p) in enumerate(zip(tiles, prm))]
    for crop_h in (None, ch):
        packed, params, ids = RawCollate(crop_h)(batch)
        assert isinstance(packed, PackedTiles) and len(packed) == len(tiles) and ids[3] == 'id3'
        if crop_h:
            assert packed[0][0].shape[0] == ch and params[0][0] == 0 and packed[2][0].shape[0] == 40      # cut to the crop's rows; a smaller tile stays whole
        got_i, got_l = aug.prepare(packed, params)
        assert torch.equal(got_i, want_i) and torch.equal(got_l, want_l)
    # pairs: (novel, base) = (tile 2i, tile 2i + 1)
    pb = [(((tiles[2 * i]), (tiles[2 * i + 1])), (prm[2 * i], prm[2 * i + 1]), 'n%d' % i) for i in range(4)]
    pa = PairAugmenter((ch, cw), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 255, device=DEV)
    w = pa.prepare([b[0] for b in pb], [b[1] for b in pb])
    for crop_h in (None, ch):
        packed, params, _ = PairCollate(crop_h)(pb)
        g = pa.prepare(packed, params)
        assert all(torch.equal(a, b) for a, b in zip(g, w))
    assert torch.equal(w[0], want_i[0::2]) and torch.equal(w[2], want_i[1::2])


# --------------------------------------------------------------------------------------------- config 4: the optimizer of ft_pop as one capturable launch
def test_sgd_kernel_equals_torch_sgd(hip):
    """segland_amd.optim.SGD (csrc/optim.hip sl_sgd_multi: every parameter in one launch) against torch.optim.SGD, the reference's fine-tuning optimizer
    (ft_pop.py:205-209: momentum 0.9, weight decay, two parameter groups with their own lr): parameters and momentum buffers to 1e-6 over five steps (torch's
    multi-tensor kernels may contract g + wd p into one fused multiply-add; this build compiles with -ffp-contract=off) with a changing learning rate, a parameter
    whose first gradient arrives late, odd sizes, and a state_dict round trip from torch's optimizer into this one."""
    from segland_amd.optim import SGD
    torch.manual_seed(1)
    shapes = [(33, 7), (513,), (64, 16, 3, 3), (4, 512), (1,)]
    mine = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    groups = lambda ps: [dict(params=ps[:3], lr=1e-2), dict(params=ps[3:], lr=1e-1, weight_decay=0.0)]
    a = SGD(groups(mine), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    b = torch.optim.SGD(groups(ref), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    for it in range(5):
        for o in (a, b):
            for gi, g in enumerate(o.param_groups):
                g['lr'] = (1e-2 if gi == 0 else 1e-1) * (1 - it / 8.0)
        for k, (p, q) in enumerate(zip(mine, ref)):
            if k == 4 and it < 2:
                p.grad = q.grad = None                      # first gradient at iteration 2
                continue
            gr = torch.randn_like(p)
            p.grad, q.grad = gr.clone(), gr.clone()
        a.step(); b.step()
        for k, (p, q) in enumerate(zip(mine, ref)):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (it, k, float((p - q).abs().max()))
        if it == 2:                                         # resume: torch's state_dict layout in, torch's out
            import copy
            sd = copy.deepcopy(b.state_dict())              # load_state_dict keeps the tensors it is given: without the copy both optimizers would share their momentum buffers
            a2 = SGD(groups(mine), lr=1e-2, momentum=0.9, weight_decay=5e-4)
            a2.load_state_dict(sd)
            a = a2
    for p, q in zip(mine, ref):
        ba, bb = a.state[p].get('momentum_buffer'), b.state[q].get('momentum_buffer')
        assert (ba is None) == (bb is None) and (ba is None or torch.allclose(ba, bb, rtol=1e-6, atol=1e-7))
    with pytest.raises(NotImplementedError):
        SGD(mine, lr=0.1, momentum=0.9, nesterov=True)


# --------------------------------------------------------------------------------------------- scratch buffers of captured calls
def test_workspace_of_a_captured_call_belongs_to_its_graph(hip):
    """ops.workspace while the stream is capturing: memory of the capturing graph's pool, never the process-wide cache entry (all captures run on torch's one
    capture stream, so that entry was shared by every graph; when it grew in the middle of a capture the nodes recorded before kept the address of the dropped
    tensor -- a block of an older, destroyed graph's pool: GPU memory fault at the first replay, gpurun r4k..r4m).  The sequence that faulted: a first graph with a
    small weight gradient dies; a second graph records a small and then a large weight gradient; its replays equal the eager results."""
    import gc
    from segland_amd import ops
    torch.manual_seed(5)
    small = (torch.randn(2, 12, 16, 64, device=DEV).to(torch.bfloat16), torch.randn(2, 12, 16, 64, device=DEV).to(torch.bfloat16), ops.ConvSpec(64, 64, 3, 1, 1, 1))
    big = (torch.randn(2, 96, 128, 64, device=DEV).to(torch.bfloat16), torch.randn(2, 96, 128, 256, device=DEV).to(torch.bfloat16), ops.ConvSpec(64, 256, 1, 1, 0, 1))
    want_s, want_b = ops.conv2d_bwd_weight(*small).clone(), ops.conv2d_bwd_weight(*big).clone()
    keys_before = set(ops._ws_cache)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        w = ops.workspace(1 << 20, torch.device(DEV, 0), 'wgrad')
        assert all(w.data_ptr() != c.data_ptr() for c in ops._ws_cache.values())
        o1 = ops.conv2d_bwd_weight(*small)
    g1.replay()
    torch.cuda.synchronize()
    assert torch.equal(o1, want_s)
    del g1, o1, w
    gc.collect()
    torch.cuda.empty_cache()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        o_s = ops.conv2d_bwd_weight(*small)
        o_b = ops.conv2d_bwd_weight(*big)
    assert set(ops._ws_cache) == keys_before, 'a capture left an entry in the eager workspace cache'
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(o_s, want_s) and torch.equal(o_b, want_b)


# --------------------------------------------------------------------------------------------- config 4: few-tile inference convs on 64-row tiles
@pytest.mark.parametrize('cin,cout,k,dil', [(1024, 256, 1, 1), (256, 256, 3, 2), (512, 128, 1, 1), (128, 128, 3, 1)])
def test_few_tile_inference_convs_on_64_row_tiles(hip, cin, cout, k, dil):
    """A frozen conv + folded BatchNorm of a fine-tune pair (8 192 rows, 128 / 256 output channels) has 64 / 128 tiles of 128 x 128 for 256 CUs; launch_gemm gives such
    launches 64 x 128 tiles (conv_gemm_ring_kernel<64, 128>: 1024 -> 256 19.0 -> 15.4 us, 3x3 256 -> 256 d2 37.4 -> 30.0; 441 -> 454 pairs/s, tools/ring64_check.py).
    1x1 layers: same K order per output element, bit-identical to the 128 x 128 tiles (hook sl_debug_ring64_max_tiles(0)), incl. a ragged last row block; all close to fp32 torch.
    Round 5: the 64-row tiles stream 128-byte stage rows (profiles/r5_ab_ring64_geom.txt: 459.5 -> 469.6 pairs/s)."""
    import torch.nn.functional as F
    from segland_amd import ops
    dt = torch.bfloat16
    g = torch.Generator(device='cpu').manual_seed(cin + cout + k)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dt).float().to(DEV)
    wf, _ = ops.weight_prep(w, dt)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.2).to(DEV)
    spec = ops.ConvSpec(cin, cout, k, 1, dil * (k // 2), dil)
    for B, H, W in ((2, 64, 64), (1, 50, 52)):           # 8 192 rows; 2 600 rows: a ragged last row block (from 2 048 rows on, like the 128 x 128 ring tiles)
        x = torch.randn(B, H, W, cin, generator=g).to(dt).to(DEV)
        rs = torch.randn(B, H, W, cout, generator=g).to(dt).to(DEV)
        try:
            hip.sl_debug_ring64_max_tiles(0)
            y128 = ops.conv2d_affine_fwd(x, wf, spec, scale, shift, residual=rs, relu=True).clone()
        finally:
            hip.sl_debug_ring64_max_tiles(256)
        y64 = ops.conv2d_affine_fwd(x, wf, spec, scale, shift, residual=rs, relu=True)
        if k == 1:
            assert torch.equal(y64, y128)
        else:          # round 5: 128-byte stage rows = 64-channel chunks; a 3x3 gather sums (chunk, tap) in another order than the 32-channel chunks of the 128 x 128 tiles
            assert float((y64.float() - y128.float()).abs().max()) <= 1e-2 * float(y128.float().abs().max())
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 1, dil * (k // 2), dil).permute(0, 2, 3, 1).to(dt).float()
        ref = torch.relu(ref * scale + shift + rs.float())
        assert float((y64.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
