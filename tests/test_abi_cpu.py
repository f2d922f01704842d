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
