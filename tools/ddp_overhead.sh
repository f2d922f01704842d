#!/bin/bash
# One-GPU cost of the data-parallel wrapper: bench.py plain, under DDP/RCCL with the in-place bucket gradients (default when
# bench.py runs under DDP; the timed steps of every line are uninstrumented), and under stock DDP (per-parameter copy + scale kernels), back to back on the SAME box.  Usage: bash tools/ddp_overhead.sh [steps]
STEPS=${1:-50}
line() { grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.1f tiles/s  %7.3f ms/step  median %7.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median']))" "$1"; }
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1
for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --no-step-graph --steps $STEPS 2>/dev/null | line "plain, kernel by kernel"
  python3 bench.py --no-cpu-baseline --steps $STEPS 2>/dev/null | line "plain, one graph per step"
  SEGLAND_FORCE_DDP=1 MASTER_PORT=2951$rep python3 bench.py --no-cpu-baseline --steps $STEPS 2>/dev/null | line "DDP, in-place bucket grads"
  SEGLAND_FORCE_DDP=1 SEGLAND_DDP_PLAIN=1 MASTER_PORT=2952$rep python3 bench.py --no-cpu-baseline --steps $STEPS 2>/dev/null | line "DDP, stock copy+scale"
done
