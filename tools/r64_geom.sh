#!/bin/bash
# 64 x 128 ring tiles with 128-byte stage rows (round 5): per-layer times against the 128 x 128 tiles, and the fine-tune pair rate by the tile-count limit of the 64-row form
python tools/ring64_check.py 2>&1 | grep -v amdgpu.ids | tail -10
for lim in 160 256 320 512; do for rep in 1 2; do
  python tools/with_hook.py sl_debug_ring64_max_tiles=$lim -- tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('limit $lim: ft pair', d['value'], d['unit'])"
done; done
for lim in 160 256; do
  python tools/with_hook.py sl_debug_ring64_max_tiles=$lim -- tools/bench_ft.py --dtype bf16 --model swin_pop 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('limit $lim: ft pair swin', d['value'], d['unit'])"
  python tools/with_hook.py sl_debug_ring64_max_tiles=$lim -- bench.py --model swin_pop --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('limit $lim: swin_pop step', d['value'], d['unit'])"
done
