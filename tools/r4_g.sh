#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
for a in "--workers 16" "--workers 32" "--workers 32 --compression tiff_lzw" "--workers 64 --compression tiff_lzw" "--workers 64 --compression tiff_adobe_deflate" "--workers 16 --source randint" "--workers 32 --source randint" \
         "--workers 16 --pairs --batch 1 --shot 5 --batches 400" "--workers 32 --pairs --batch 16 --shot 40 --batches 30"; do
  echo "== feed_rate.py $a"; timeout 400 python tools/feed_rate.py $a 2>&1 | tail -1; done > $O/feed_rate.txt
timeout 900 python -m pytest tests/test_round3_gpu.py -k "two_ranks or world1" tests/test_round2_gpu.py::test_two_ranks_equal_one_full_batch tests/test_round2_gpu.py::test_argmax_and_pseudo_labels_vs_same_box_oracle tests/test_round4_gpu.py -m gpu -q -x --durations=10 > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt
bash tools/r4_pmc.sh > $O/pmc.log 2>&1
cp -r $R/gpurun_out/r4pmc/* $O/ 2>/dev/null
