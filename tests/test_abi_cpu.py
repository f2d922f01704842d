"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/segland_hip.h declares.
No compute calls (there is no GPU here); argument validation paths that return before any launch are exercised."""
import ctypes as C
import os

import pytest

from segland_amd import _lib


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    decl = _lib.declared_functions()
    assert len(decl) >= 45
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.sl_version() >= 100


def test_no_exported_symbol_is_undeclared(lib):
    """Every `sl_*` symbol the library exports is declared in the product header or in the debug header (include/segland_hip_debug.h: test / tuning hooks, which a
    deployment never calls); every hook of the debug header exists, and sl_debug_reset() is callable without a GPU."""
    import subprocess
    names = subprocess.run('nm -D --defined-only %s' % _lib.LIB_PATH, shell=True, capture_output=True, text=True, check=True).stdout.split('\n')
    exported = {l.split()[-1] for l in names if ' T sl_' in l}
    prod, dbg = _lib.declared_functions(), _lib.declared_functions(_lib.DEBUG_HEADER)
    assert exported and not (exported - set(prod) - set(dbg)), sorted(exported - set(prod) - set(dbg))
    assert all(n.startswith('sl_debug_') for n in dbg) and not any(n.startswith('sl_debug_') for n in prod)
    for n in dbg:
        assert hasattr(lib, n), n
    lib.sl_debug_reset()


def test_bad_descriptor_is_rejected_without_launch(lib):
    d = _lib.SlConvDesc(_lib.SL_BF16, 1, 8, 8, 48, 64, 1, 1, 1, 0, 1, 8, 8, 48)   # Cin not a multiple of 64
    dummy = C.c_void_p(16)
    rc = lib.sl_conv2d_fwd(C.byref(d), dummy, None, dummy, None, 0, dummy, None, None)
    assert rc == -1 and b'multiples of' in lib.sl_last_error_string()
    d2 = _lib.SlConvDesc(_lib.SL_F32, 1, 8, 8, 64, 64, 3, 3, 1, 1, 1, 7, 8, 64)   # inconsistent Ho
    assert lib.sl_conv2d_fwd(C.byref(d2), dummy, None, dummy, None, 0, dummy, None, None) == -1
    assert lib.sl_conv2d_stat_rows(C.byref(_lib.SlConvDesc(0, 2, 16, 16, 64, 64, 1, 1, 1, 0, 1, 16, 16, 64))) == 4


def test_product_ops_refuse_cpu_tensors():
    import torch
    from segland_amd import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.bn_act(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))


def test_more_argument_validation_without_launch(lib):
    """Every entry point validates before it touches the device: negative SL_E* codes + a message, no launch (runs without a GPU)."""
    dummy = C.c_void_p(16)
    d = _lib.SlConvDesc(_lib.SL_BF16, 2, 16, 16, 128, 128, 3, 3, 1, 1, 1, 16, 16, 128)
    need = lib.sl_conv2d_bwd_weight_workspace(C.byref(d))
    assert need > 0
    assert lib.sl_conv2d_bwd_weight(C.byref(d), dummy, None, dummy, dummy, dummy, need - 1, None) == -2          # SL_EWORKSPACE
    assert b'workspace' in lib.sl_last_error_string()
    assert lib.sl_conv2d_bwd_weight_ex(C.byref(d), dummy, None, dummy, dummy, 64, 0, dummy, need, None) == -1  # window narrower than Cin
    p = _lib.SlPpmDesc(_lib.SL_BF16, 2, 16, 16, 128, 5, (C.c_int * 4)(1, 2, 3, 6))                              # 5 levels
    assert lib.sl_ppm_workspace(C.byref(p)) == 0
    p4 = _lib.SlPpmDesc(_lib.SL_BF16, 2, 16, 16, 128, 4, (C.c_int * 4)(1, 2, 3, 6))
    assert lib.sl_ppm_workspace(C.byref(p4)) > 0
    assert lib.sl_ppm_pool_fwd(C.byref(p4), dummy, dummy, dummy, 16, None) == -2
    assert lib.sl_ppm_rows_gemm(C.byref(p4), 48, 64, dummy, dummy, dummy, None, None, 0, None) == -1             # K % 32
    assert lib.sl_ppm_rows_gemm_stat_rows(C.byref(p4)) == 4                                                     # ceil(2*s*s/128) per level
    assert lib.sl_confusion_matrix(dummy, dummy, 100, 65, 255, dummy, None) == -1                               # K > 64
    assert lib.sl_upsample_ce_fwd is not None and lib.sl_weight_prep_batched(None, 1, 1, None) == -1


def test_kernel_dispatch_and_statistic_rows_by_shape(lib):
    """Which kernel a conv shape runs on (sl_conv2d_tile_config: 1000000 x family + 1000 x rows + columns, what bench.py attributes its HIP-event
    timings with) and the granularity of the BN statistic partials it writes -- host logic, no launch.  Family 5 = half-tile kernel
    (conv_gemm_p8_kernel), 6 = pixel-stationary short-K kernel, 7 = 64 -> 64 3x3 patch kernel, 8 = 3x3 patch kernel of the wide layers."""
    bf = _lib.SL_BF16

    def desc(B, H, W, cin, cout, k, pad=0, dil=1, dtype=bf):
        return _lib.SlConvDesc(dtype, B, H, W, cin, cout, k, k, 1, pad, dil, H, W, cin)
    # layer3 conv3 (256 -> 1024) forward and layer3 conv1 (1024 -> 256) data gradient: K = 256 at 65 536 pixels -> stationary kernel, one partial row per 256 pixels
    d = desc(16, 64, 64, 256, 1024, 1)
    assert lib.sl_conv2d_tile_config(C.byref(d), 0) == 6256064 and lib.sl_conv2d_stat_rows(C.byref(d)) == 256
    assert lib.sl_conv2d_tile_config(C.byref(desc(16, 64, 64, 1024, 256, 1)), 1) == 6256064
    # the same layers the other way round reduce over 1024 channels: half-tile kernel
    assert lib.sl_conv2d_tile_config(C.byref(d), 1) == 5256256
    assert lib.sl_conv2d_tile_config(C.byref(desc(16, 64, 64, 1024, 256, 1)), 0) == 5256256
    # below 65 536 pixels, in fp32, or with a 3x3 window the stationary kernel does not apply
    assert lib.sl_conv2d_tile_config(C.byref(desc(8, 64, 64, 256, 1024, 1)), 0) == 5256256
    assert lib.sl_conv2d_tile_config(C.byref(desc(16, 64, 64, 256, 1024, 1, dtype=_lib.SL_F32)), 0) // 1000000 == 4
    # 3x3 stride-1 layers of the dilated trunk at >= 32 768 pixels, map a multiple of 16: patch kernel (family 8); a ragged map or a small batch: the tile kernels
    assert lib.sl_conv2d_tile_config(C.byref(desc(16, 64, 64, 256, 256, 3, pad=2, dil=2)), 0) == 8256256
    assert lib.sl_conv2d_tile_config(C.byref(desc(16, 64, 64, 512, 512, 3, pad=4, dil=4)), 1) == 8256256
    assert lib.sl_conv2d_tile_config(C.byref(desc(18, 60, 64, 256, 256, 3, pad=2, dil=2)), 0) == 5256256
    assert lib.sl_conv2d_tile_config(C.byref(desc(8, 64, 64, 256, 256, 3, pad=2, dil=2)), 0) == 8256256          # 32 768 pixels: still the patch kernel
    assert lib.sl_conv2d_tile_config(C.byref(desc(4, 64, 64, 256, 256, 3, pad=2, dil=2)), 0) // 1000000 != 8
    # layer1.conv2: 64 -> 64 3x3 at 128 x 128 -> patch kernel, one partial row per 16 x 16-pixel tile; a ragged map rounds the tile grid up
    d = desc(16, 128, 128, 64, 64, 3, pad=1)
    assert lib.sl_conv2d_tile_config(C.byref(d), 0) == 7016016 and lib.sl_conv2d_tile_config(C.byref(d), 1) == 7016016
    assert lib.sl_conv2d_stat_rows(C.byref(d)) == 16 * 8 * 8
    assert lib.sl_conv2d_stat_rows(C.byref(desc(5, 120, 136, 64, 64, 3, pad=1))) == 5 * 8 * 9
    # a small map of the same layer stays on the tile kernels (128-row blocks below 24 576 pixels)
    d = desc(2, 64, 64, 64, 64, 3, pad=1)
    assert lib.sl_conv2d_tile_config(C.byref(d), 0) // 1000000 == 2 and lib.sl_conv2d_stat_rows(C.byref(d)) == 2 * 64 * 64 // 128


def _desc(dtype, B, H, W, cin, cout, k, stride, pad, dil):
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    return _lib.SlConvDesc(dtype, B, H, W, cin, cout, k, k, stride, pad, dil, Ho, Wo, cin)


def test_dispatch_queries_answer_from_the_dispatch_itself(lib):
    """Round 5: sl_conv2d_tile_config(_ex) and sl_conv2d_wgrad_config are the predicate chains the launches switch on (host code only: runs without a GPU).
    ResNet-50 bench shapes (B 16, 64 x 64 / 128 x 128): which kernel, by epilogue."""
    STATS, AFFINE, ADDEND, BITS, GATE = 1, 2, 4, 8, 16
    fam = lambda d, mode, epi: lib.sl_conv2d_tile_config_ex(C.byref(d), mode, epi) // 1000000
    d = _desc(_lib.SL_BF16, 16, 64, 64, 512, 2048, 1, 1, 0, 1)
    assert lib.sl_conv2d_tile_config_ex(C.byref(d), 0, STATS) == 5256256 and lib.sl_conv2d_tile_config(C.byref(d), 0) == 5256256      # half-tile kernel (the K = 512 pixel-stationary form is off by default)
    d = _desc(_lib.SL_BF16, 16, 64, 64, 256, 1024, 1, 1, 0, 1)
    assert fam(d, 0, STATS) == 6 and fam(d, 0, AFFINE) == 5                  # pixel-stationary for the training form, half-tile for the folded-BN inference form
    assert fam(d, 1, GATE) == 5                                               # conv3's data gradient (K = 1024) with the gated-statistics store phase
    d = _desc(_lib.SL_BF16, 16, 64, 64, 1024, 256, 1, 1, 0, 1)
    assert fam(d, 1, ADDEND | GATE) == 6 and fam(d, 1, ADDEND | BITS) == 6    # conv1's data gradient (K = 256): MODE 5 / MODE 2 of the pixel-stationary kernel
    assert lib.sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(d)) == 16 * 64 * 64 // 256
    d = _desc(_lib.SL_BF16, 16, 64, 64, 2048, 512, 1, 1, 0, 1)
    assert lib.sl_conv2d_bwd_data_addend_bnstat_rows(C.byref(d)) == 0          # K = 512: the half-tile kernel has no cross-block store phase (profiles/r5_ab_bn_fusions.txt)
    d = _desc(_lib.SL_BF16, 16, 64, 64, 512, 512, 3, 1, 4, 4)
    assert fam(d, 0, STATS) == 8 and fam(d, 1, GATE) == 8 and lib.sl_conv2d_wgrad_config(C.byref(d)) == 3      # 3x3 patch kernel both ways, nine-tap weight gradient
    d = _desc(_lib.SL_BF16, 2, 64, 64, 2048, 512, 3, 1, 1, 1)
    assert lib.sl_conv2d_tile_config_ex(C.byref(d), 0, AFFINE | 32) >= 10000000      # the fine-tune pair's pyramid conv: split-K on the patch kernel
    d = _desc(_lib.SL_BF16, 16, 128, 128, 128, 128, 3, 2, 1, 1)
    assert lib.sl_conv2d_wgrad_config(C.byref(d)) != 3                       # stride 2: per-tap kernels
    d = _desc(_lib.SL_F32, 16, 64, 64, 256, 256, 3, 1, 2, 2)
    assert lib.sl_conv2d_wgrad_config(C.byref(d)) != 3                       # fp32 parity mode: per-tap kernels
    out = (C.c_int * 8)()
    for shape, blocks in (((16, 64, 64, 2048, 512, 1), 256), ((16, 64, 64, 512, 512, 4), 256), ((16, 64, 64, 256, 256, 2), 256), ((16, 64, 64, 128, 128, 1), 256)):
        B, H, W, cin, cout, dil = shape
        d = _desc(_lib.SL_BF16, B, H, W, cin, cout, 3, 1, dil, dil)
        assert lib.sl_debug_wgrad3_plan(C.byref(d), out) == 1 and out[7] == blocks, (shape, list(out))
        served, ppu, L, SP, ppb, splits, tiles, nblk = list(out)
        units = B * dil * dil * (W // dil // 16)
        assert tiles == (cout // 128) * (cin // 64) and splits * ppb >= units * ppu and SP == L + (1 if ppu == 1 else 2) and L * ppu >= H // dil
        assert lib.sl_conv2d_bwd_weight_workspace(C.byref(d)) >= nblk * 8 * 4 * 9 * 64 * 4 * 4
