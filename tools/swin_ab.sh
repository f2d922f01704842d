#!/bin/bash
# Swin-T POP step, same box, A/B/A/B of one environment switch:  bash tools/swin_ab.sh SEGLAND_SWIN_GELU_FUSE
V=$1
for rep in 1 2; do for v in 1 0; do
  echo -n "$V=$v swin_pop: "; env $V=$v python bench.py --model swin_pop --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done; done
