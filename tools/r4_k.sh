#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_swin_gpu.py tests/test_kernels_gpu.py tests/test_round3_gpu.py -k "not two_ranks and not world1" -m gpu -q -x --durations=5 > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt
python bench.py --model swin_pop 2>/dev/null | grep '^{"metric"' > $O/r4_bench_swin.json
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $O/bench_default.json
