#!/bin/bash
# round 4: after the workspace-under-capture fix -- the order that faulted, then the whole GPU suite in reverse file order and in the driver's order
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_swin_gpu.py tests/test_round3_gpu.py -k "not two_ranks and not world1" -m gpu -q -x > $O/pytest_bad_order.txt 2>&1; echo "rc $?" >> $O/pytest_bad_order.txt
SEGLAND_PARITY_LOG=$O/parity_log.txt timeout 1500 python -m pytest tests/ -m gpu -q -x --durations=25 > $O/pytest_gpu_full.txt 2>&1; echo "rc $?" >> $O/pytest_gpu_full.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
