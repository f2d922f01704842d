#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
for a in "--workers 16" "--workers 32" "--workers 16 --compression tiff_lzw" "--workers 32 --compression tiff_lzw" "--workers 32 --compression tiff_adobe_deflate" "--workers 16 --source randint" \
         "--workers 16 --pairs --batch 1 --shot 5 --batches 400" "--workers 32 --pairs --batch 1 --shot 40 --batches 400" "--workers 32 --pairs --batch 16 --shot 40 --batches 30" "--workers 32 --pairs --batch 16 --shot 40 --batches 30 --compression tiff_lzw"; do
  echo "== feed_rate.py $a"; timeout 300 python tools/feed_rate.py $a 2>&1 | tail -1; done > $O/feed_rate.txt
for v in 0 1 2 3; do echo "== ring128 cfg $v"; python tools/ft_shapes.py --ring128 $v 2>&1 | grep -v amdgpu.ids | grep "4128128\|sum over"; done > $O/ft_shapes_ring.txt
for v in 0 1 2 3; do echo "== ring128 cfg $v"; python tools/with_hook.py sl_debug_conv_ring128=$v -- bench.py --model swin_pop --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  python tools/with_hook.py sl_debug_conv_ring128=$v -- tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{'; done > $O/ring_ab.txt
timeout 600 python -m cProfile -o /tmp/ftprof.out -m pytest tests/test_round2_gpu.py::test_ft_pop_with_loader_workers_and_pinned_memory -q -x > $O/slow_test.txt 2>&1
python -c "
import pstats; p=pstats.Stats('/tmp/ftprof.out'); p.sort_stats('cumulative').print_stats(45)" >> $O/slow_test.txt 2>&1
timeout 900 python -m pytest tests/test_round2_gpu.py -k "raw_tiles_with_workers or raw_pairs_with_workers or g17 or g19 or eval_base_on_raw" -m gpu -q -x --durations=10 > $O/pytest_feed.txt 2>&1; echo "rc $?" >> $O/pytest_feed.txt
