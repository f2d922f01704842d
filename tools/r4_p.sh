#!/bin/bash
# round 4: 64 x 128 ring tiles (default from 160 tiles down): the Swin training step and the fine-tune steps with the hook at 0 / 160 / 200; tests that touch small inference convs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
for t in 0 160 200; do
  echo "== sl_debug_ring64_max_tiles=$t" >> $O/ring64_swin.txt
  python tools/with_hook.py sl_debug_ring64_max_tiles=$t -- bench.py --model swin_pop --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/ring64_swin.txt
  python tools/with_hook.py sl_debug_ring64_max_tiles=$t -- tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' | cut -c1-140 >> $O/ring64_swin.txt
done
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_swin_gpu.py tests/test_round4_gpu.py tests/test_drivers_gpu.py -m gpu -q -x > $O/pytest_subset.txt 2>&1; echo "rc $?" >> $O/pytest_subset.txt
