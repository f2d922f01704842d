#!/bin/bash
# Same-box A/B/A/B of bench.py under environment switches.  usage: bash tools/ab_bench.sh "<label>=<ENV=..>" ... ; first entry is the baseline.
# e.g. bash tools/ab_bench.sh "default=" "no_fork=SEGLAND_WGRAD_REDUCE_STREAM=0"
STEPS=${STEPS:-40}
line() { grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %8.1f tiles/s  %7.3f ms/step  median %7.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_median']))" "$1"; }
for rep in 1 2; do
  for ent in "$@"; do
    label=${ent%%=*}; envs=${ent#*=}
    env $envs SEGLAND_BENCH_NOEVENTS=1 python3 bench.py --no-cpu-baseline --steps $STEPS $BENCH_ARGS 2>/dev/null | line "$label"
  done
done
