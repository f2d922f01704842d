#!/bin/bash
# Same-box A/B of two COMMITS (library + Python side): prepares ./_ab/<rev>/ (a `git worktree`-like export of the revision with its library built) HERE, then on the GPU box
#   bash tools/ab_commits.sh run <revA> <revB> [bench.py args...]        alternates `python bench.py` from both trees (A B A B) and prints tiles/s, ms per step
# usage here:  bash tools/ab_commits.sh prepare <rev> [<rev> ...]        (HEAD~1, a sha, ...; the working tree itself is ".")
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cmd=$1; shift
if [ "$cmd" = prepare ]; then
  for rev in "$@"; do
    sha=$(git -C $R rev-parse --short $rev)
    d=$R/_ab/$sha
    rm -rf $d; mkdir -p $d
    git -C $R archive $rev segland_amd include oracle bench.py | tar -x -C $d
    make -C $d/segland_amd/csrc -j8 > $d/build.log 2>&1 || { echo "build of $rev failed"; tail -5 $d/build.log; exit 1; }
    rm -f $d/segland_amd/csrc/*.o
    echo "prepared $rev -> _ab/$sha"
  done
elif [ "$cmd" = run ]; then
  A=$1; B=$2; shift 2
  dir() { if [ "$1" = . ]; then echo $R; else echo $R/_ab/$1; fi; }
  for i in 1 2 3; do
    for t in $A $B; do
      ( cd $(dir $t) && python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', j['value'], j['unit'], j['ms_per_step'], 'ms/step')" )
    done
  done
fi
