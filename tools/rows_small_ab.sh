#!/bin/bash
# conv_rows_small_kernel (<= 32-row launches: the prototype rows of the POP head's MLP) on / off: fine-tune pair and ResNet-50 step, same box
V=SEGLAND_CONV_ROWS_SMALL
for rep in 1 2; do for v in 1 0; do
  echo -n "$V=$v ft pair: "; env $V=$v python tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done; done
for v in 1 0; do
  echo -n "$V=$v ft pair swin: "; env $V=$v python tools/bench_ft.py --dtype bf16 --model swin_pop 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
  echo -n "$V=$v resnet50: "; env $V=$v python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"
done
