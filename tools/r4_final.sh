#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r4_parity_log.txt
SEGLAND_PARITY_LOG=$O/r4_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q -x --durations=30 > $O/r4_pytest_durations.txt 2>&1; echo "rc $?" >> $O/r4_pytest_durations.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r4_smoke.txt 2>&1
bash tools/collect_profiles.sh > $O/r4_collect.log 2>&1
