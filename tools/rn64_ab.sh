#!/bin/bash
# 128 x 64 ring tiles for 64-column launches without BN statistics (SEGLAND_CONV_RINGN64): fine-tune pair, Swin-T step, ResNet-50 step, same box A/B/A/B
V=SEGLAND_CONV_RINGN64
for rep in 1 2; do for v in 1 0; do
  echo -n "$V=$v ft pair: "; env $V=$v python tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done; done
for rep in 1 2; do for v in 1 0; do
  echo -n "$V=$v swin_pop: "; env $V=$v python bench.py --model swin_pop --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done; done
for v in 1 0; do
  echo -n "$V=$v ft pair swin: "; env $V=$v python tools/bench_ft.py --dtype bf16 --model swin_pop 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
  echo -n "$V=$v resnet50: "; env $V=$v python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done
