#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r4_parity_log.txt
SEGLAND_PARITY_LOG=$O/r4_parity_log.txt timeout 1500 python -m pytest tests/ -m gpu -q -x --durations=25 > $O/r4_pytest_gpu_full.txt 2>&1; echo "rc $?" >> $O/r4_pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r4_smoke.txt 2>&1
python bench.py > $O/r4_bench_last.json 2> $O/bench_err.txt
