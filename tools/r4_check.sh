#!/bin/bash
# round-4 quick check: a subset of the GPU suite + the bench line + per-kernel stats.  usage: bash tools/r4_check.sh <tag> [pytest args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r4b}; shift || true
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest "$@" -m gpu -q -x --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
python bench.py --no-cpu-baseline 2>$O/bench_err.txt | grep '^{"metric"' > $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
