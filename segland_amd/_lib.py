"""ctypes binding of libsegland_hip.so (the C ABI declared in include/segland_hip.h).

The product path has no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os
import re
import subprocess

import torch  # noqa: F401  -- must be imported BEFORE libsegland_hip.so so both share torch's HIP runtime (libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('SEGLAND_LIB_PATH') or os.path.join(CSRC, 'libsegland_hip.so')      # SEGLAND_LIB_PATH: a differently built library (kernel A/B from one checkout)
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'segland_hip.h')
DEBUG_HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'segland_hip_debug.h')      # test / tuning hooks (sl_debug_*): not part of the product ABI

SL_F32, SL_BF16 = 0, 1


class SlConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('dtype', 'B', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'pad', 'dil', 'Ho', 'Wo', 'C1')]


class SlPpmDesc(C.Structure):
    _fields_ = [('dtype', C.c_int), ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('C', C.c_int), ('nlevels', C.c_int),
                ('sizes', C.c_int * 4)]


class SlWinDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('dtype', 'B', 'H', 'W', 'C', 'heads', 'qkv_pitch', 'out_pitch', 'shift')]


class SlResizeDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('dtype', 'B', 'h', 'w', 'H', 'W', 'C', 'src_pitch', 'src_off', 'dst_pitch', 'dst_off', 'align_corners', 'accumulate', 'src_f32')]


SL_NORM_MAX = 64


class SlNormBatch(C.Structure):
    _fields_ = [('grad', C.c_void_p * SL_NORM_MAX), ('numel', C.c_longlong * SL_NORM_MAX), ('chunk0', C.c_int * (SL_NORM_MAX + 1)), ('n', C.c_int), ('chunk_base', C.c_int)]


SL_WGRAD_BATCH_MAX = 8


class SlWgradReduce(C.Structure):
    _fields_ = ([('ws', C.c_void_p), ('dw', C.c_void_p), ('total', C.c_longlong)]
                + [(n, C.c_int) for n in ('splits', 'Cin', 'dw_cin_total', 'dw_ci_off', 'n_valid', 'c_valid', 'dtype', 'Cout', 'ncol')]
                + [('dy', C.c_void_p), ('rows', C.c_longlong), ('rows_per_block', C.c_longlong), ('colsum_part', C.c_void_p)])


SL_COLSUM_MAX = 12


class SlColsumBatch(C.Structure):
    _fields_ = [('n', C.c_int), ('part', C.c_void_p * SL_COLSUM_MAX), ('out', C.c_void_p * SL_COLSUM_MAX), ('nblk', C.c_int * SL_COLSUM_MAX),
                ('C', C.c_int * SL_COLSUM_MAX)]


_CTYPE = {
    'int': C.c_int, 'long long': C.c_longlong, 'float': C.c_float, 'size_t': C.c_size_t, 'sl_stream_t': C.c_void_p,
}


def _arg_ctype(decl):
    decl = decl.strip()
    if '*' in decl:
        return C.c_void_p
    base = re.sub(r'\b\w+$', '', decl).replace('const', '').strip()
    return _CTYPE[base]


def declared_functions(header=None):
    """{name: (restype, [argtypes])} parsed from the header (default: the product ABI, include/segland_hip.h), so the binding cannot drift from the ABI."""
    src = open(header or HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'(int|size_t|void|const char\s*\*)\s+(sl_\w+)\s*\(([^;{]*?)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = [] if args in ('', 'void') else [_arg_ctype(a) for a in args.split(',')]
        restype = {'int': C.c_int, 'size_t': C.c_size_t, 'void': None}.get(ret, C.c_char_p)
        out[name] = (restype, argtypes)
    return out


def build(force=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(['make', '-C', CSRC, 'clean'], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(['make', '-C', CSRC, '-j8'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('building libsegland_hip.so failed:\n' + r.stdout[-4000:])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libsegland_hip.so is missing (%s): run `python -c "import __graft_entry__ as g; g.build()"` '
                               'or `make -C segland_amd/csrc`; there is no fallback path' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for hdr in (HEADER, DEBUG_HEADER):
            for name, (restype, argtypes) in declared_functions(hdr).items():
                fn = getattr(l, name)            # AttributeError here == header/library mismatch
                fn.restype, fn.argtypes = restype, argtypes
        _lib = l
    return _lib


def check(code, what=''):
    if code != 0:
        msg = lib().sl_last_error_string()
        raise RuntimeError('segland_hip %s failed (%d): %s' % (what, code, msg.decode() if msg else '?'))
