#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_round3_gpu.py -k "backward_cut" -m gpu -q -x > $O/pytest_cut_alone.txt 2>&1; echo "rc $?" >> $O/pytest_cut_alone.txt
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_graph_step_gpu.py -m gpu -q -x --durations=5 > $O/pytest_r4.txt 2>&1; echo "rc $?" >> $O/pytest_r4.txt
for a in "--dtype bf16" "--dtype bf16 --torch-sgd" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/ft.txt
timeout 1200 python -m pytest tests/test_swin_gpu.py tests/test_kernels_gpu.py tests/test_round3_gpu.py -k "not two_ranks and not world1" -m gpu -q -x > $O/pytest_same_as_r4k.txt 2>&1; echo "rc $?" >> $O/pytest_same_as_r4k.txt
