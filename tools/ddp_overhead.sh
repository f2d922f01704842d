#!/bin/bash
# One-GPU cost of the data-parallel wrappers: bench.py plain (one graph per step), the bucket step (two graphs around RCCL all-reduces of the build's own
# gradient buckets: the N > 1 default since round 3; at N > 1 the ResNet backward is cut -- three graphs --, at world size 1 it is not since round 4: both are run), DistributedDataParallel with in-place bucket gradients (round 2's N > 1 path, SEGLAND_BUCKET_STEP=0) and
# stock DistributedDataParallel (per-parameter copy + scale kernels), back to back on the SAME box.  Usage: bash tools/ddp_overhead.sh [steps] [extra bench args]
STEPS=${1:-50}; shift
line() { grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.1f tiles/s  %7.3f ms/step  median %7.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median']))" "$1"; }
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1
for rep in 1 2; do
  python3 bench.py --no-cpu-baseline --no-other-configs --no-step-graph --steps $STEPS "$@" 2>/dev/null | line "plain, kernel by kernel"
  python3 bench.py --no-cpu-baseline --no-other-configs --steps $STEPS "$@" 2>/dev/null | line "plain, one graph per step"
  SEGLAND_FORCE_DDP=1 SEGLAND_BUCKET_CUT=1 MASTER_PORT=2950$rep python3 bench.py --no-cpu-baseline --no-other-configs --steps $STEPS "$@" 2>/dev/null | line "RCCL world 1, bucket step with the backward cut"
  SEGLAND_FORCE_DDP=1 SEGLAND_BUCKET_CUT=0 MASTER_PORT=2953$rep python3 bench.py --no-cpu-baseline --no-other-configs --steps $STEPS "$@" 2>/dev/null | line "RCCL world 1, bucket step, no cut (world-1 default)"
  SEGLAND_FORCE_DDP=1 SEGLAND_BUCKET_STEP=0 MASTER_PORT=2951$rep python3 bench.py --no-cpu-baseline --no-other-configs --steps $STEPS "$@" 2>/dev/null | line "RCCL world 1, DDP in-place bucket grads"
  SEGLAND_FORCE_DDP=1 SEGLAND_BUCKET_STEP=0 SEGLAND_DDP_PLAIN=1 MASTER_PORT=2952$rep python3 bench.py --no-cpu-baseline --no-other-configs --steps $STEPS "$@" 2>/dev/null | line "RCCL world 1, DDP stock copy+scale"
done
