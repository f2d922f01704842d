#!/bin/bash
# round 4: the GPU suite with its files in REVERSE order and in a shuffled order (the driver runs them alphabetically; the workspace fault of this round only showed in another order)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4order; mkdir -p $O; cd $R
REV=$(ls tests/test_*gpu*.py | sort -r | tr '\n' ' ')
timeout 1500 python -m pytest $REV -m gpu -q -x -p no:cacheprovider > $O/pytest_reverse.txt 2>&1; echo "rc $?" >> $O/pytest_reverse.txt
SHUF="tests/test_round3_gpu.py tests/test_swin_gpu.py tests/test_drivers_gpu.py tests/test_round4_gpu.py tests/test_kernels_gpu.py tests/test_graph_step_gpu.py tests/test_round2_gpu.py tests/test_model_gpu.py"
timeout 1500 python -m pytest $SHUF -m gpu -q -x -p no:cacheprovider > $O/pytest_shuffled.txt 2>&1; echo "rc $?" >> $O/pytest_shuffled.txt
