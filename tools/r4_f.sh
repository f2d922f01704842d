#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
rm -f $O/parity_log.txt
SEGLAND_PARITY_LOG=$O/parity_log.txt timeout 1700 python -m pytest tests -m gpu -q -x --durations=40 > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt
for a in "--workers 16" "--workers 32" "--workers 32 --compression tiff_lzw" "--workers 64 --compression tiff_lzw" "--workers 16 --source randint" \
         "--workers 16 --pairs --batch 1 --shot 5 --batches 400" "--workers 32 --pairs --batch 1 --shot 40 --batches 400" "--workers 32 --pairs --batch 16 --shot 40 --batches 30" "--workers 64 --pairs --batch 16 --shot 40 --batches 30 --compression tiff_lzw"; do
  echo "== feed_rate.py $a"; timeout 300 python tools/feed_rate.py $a 2>&1 | tail -1; done > $O/feed_rate.txt
