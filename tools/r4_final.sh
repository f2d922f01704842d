#!/bin/bash
# round 4, final evidence run on one MI355X box: the whole GPU suite (with the parity log), smoke(), then tools/collect_profiles.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r4_parity_log.txt
SEGLAND_PARITY_LOG=$O/r4_parity_log.txt timeout 1500 python -m pytest tests/ -m gpu -q -x --durations=25 > $O/r4_pytest_gpu_full.txt 2>&1; echo "rc $?" >> $O/r4_pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r4_smoke.txt 2>&1
bash tools/collect_profiles.sh > $O/r4_collect.log 2>&1
