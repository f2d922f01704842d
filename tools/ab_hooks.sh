#!/bin/bash
# Same-box A/B of Python-side / library test hooks: alternates `bench.py` with and without the given hook settings (tools/with_hook.py) and prints the value and ms per step.
#   bash tools/ab_hooks.sh <reps> "hook=0 other.hook=0" [bench args...]
N=$1; H=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
line() { grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in $(seq $N); do
  echo -n "default: "; python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | line
  echo -n "$H: "; python $R/tools/with_hook.py $H -- bench.py --no-cpu-baseline "$@" 2>/dev/null | line
done
