#!/bin/bash
# fine-tune pair (tools/bench_ft.py) and Swin-T step, same box, A/B/A/B of one environment switch:  bash tools/ft_ab.sh SEGLAND_CONV_P8_AFFINE
V=$1
for rep in 1 2; do for v in 1 0; do
  echo -n "$V=$v ft pair: "; env $V=$v python tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done; done
for v in 1 0; do
  echo -n "$V=$v ft pair swin: "; env $V=$v python tools/bench_ft.py --dtype bf16 --model swin_pop 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
  echo -n "$V=$v swin_pop: "; env $V=$v python bench.py --model swin_pop --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done
for s in "32768 192 768" "32768 64 256" "32768 128 512"; do set -- $s
  for v in 1 0; do echo -n "$V=$v "; env $V=$v python tools/gemm_time.py --tokens $1 --cin $2 --cout $3 --gelu 2>&1 | grep "^M="; done
done
