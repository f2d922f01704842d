#!/bin/bash
# Same-box A/B/A/B of ONE environment switch on the bench lines:   bash tools/ab_env.sh SEGLAND_WGRAD_BIAS swin [r50] [ft] [ft_swin]
# (the per-switch A/B files of round 5, profiles/r5_ab_*.txt, are this loop)
V=$1; shift
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d.get('ms_per_step'), 'ms/step')"; }
one() {
  case $1 in
    r50)     python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | val ;;
    swin)    python bench.py --model swin_pop --no-cpu-baseline 2>/dev/null | grep '^{' | val ;;
    ft)      python tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' | val ;;
    ft_swin) python tools/bench_ft.py --dtype bf16 --model swin_pop 2>/dev/null | grep '^{' | val ;;
    *) echo "unknown target $1"; exit 2 ;;
  esac
}
for t in "$@"; do for rep in 1 2; do for v in 1 0; do echo -n "$V=$v $t: "; ( export $V=$v; one $t ); done; done; done
