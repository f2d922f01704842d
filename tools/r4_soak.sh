#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4soak; mkdir -p $O; cd $R
timeout 1500 python tools/soak.py > $O/soak.txt 2>&1; echo "rc $?" >> $O/soak.txt
timeout 900 python -m pytest tests/test_drivers_gpu.py tests/test_round2_gpu.py -k "workers or drivers or raw or entry or ft_pop or eval" -m gpu -q -x > $O/pytest_drivers.txt 2>&1; echo "rc $?" >> $O/pytest_drivers.txt
