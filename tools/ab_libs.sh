#!/bin/bash
# Same-box A/B of two (or more) BUILDS of the library from one checkout (SEGLAND_LIB_PATH): alternates `python bench.py` and prints tiles/s, ms per step and the per-family
# milliseconds of the instrumented step whose name contains <pattern>.   bash tools/ab_libs.sh <pattern> <reps> libsegland_hip.so libsegland_variant.so [bench args...]
P=$1; N=$2; A=$3; B=$4; shift 4
R=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $N); do for l in $A $B; do
  echo -n "$l: "; SEGLAND_LIB_PATH=$R/segland_amd/csrc/$l python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['families']
print(d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in f.items() if '$P' in k})"
done; done
