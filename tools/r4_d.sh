#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q -x --durations=10 > $O/pytest_r4.txt 2>&1; echo "rc $?" >> $O/pytest_r4.txt
python tools/loader_start.py > $O/loader_start.txt 2>&1
python tools/ft_shapes.py > $O/ft_shapes.txt 2>&1
for a in "--dtype bf16" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/ft.txt
for a in "--workers 16" "--workers 32" "--workers 16 --compression tiff_lzw" "--workers 32 --compression tiff_adobe_deflate" "--workers 16 --source randint" \
         "--workers 16 --pairs --batch 1 --shot 5 --batches 400" "--workers 32 --pairs --batch 1 --shot 40 --batches 400" "--workers 32 --pairs --batch 16 --shot 40 --batches 30" "--workers 16 --pairs --batch 16 --shot 5 --batches 10"; do
  echo "== feed_rate.py $a"; timeout 300 python tools/feed_rate.py $a 2>&1 | tail -1; done > $O/feed_rate.txt
python bench.py 2>$O/bench_err.txt | grep '^{"metric"' > $O/bench.json
