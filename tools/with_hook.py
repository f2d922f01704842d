#!/usr/bin/env python3
"""Run a script of this repository with tuning / test hooks of libsegland_hip.so set first (the hooks are not environment switches: the product has one dispatch).
usage: python tools/with_hook.py sl_debug_conv_ring192=0 [more=...] -- bench.py --model swin_pop --no-cpu-baseline
A key with dots is a module attribute of the Python side (its test hooks): segland_amd.networks.pspnet_pop._FT_TWO_BRANCH=0"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
i = sys.argv.index('--')
from segland_amd import _lib
L = _lib.lib()
for kv in sys.argv[1:i]:
    k, v = kv.split('=')
    if '.' in k:
        import importlib
        mod, attr = k.rsplit('.', 1)
        old = getattr(importlib.import_module(mod), attr)
        setattr(importlib.import_module(mod), attr, type(old)(int(v)))
    else:
        getattr(L, k)(int(v))
script = sys.argv[i + 1]
sys.argv = [script] + sys.argv[i + 2:]
runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name='__main__')
