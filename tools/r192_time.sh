#!/bin/bash
# 128 x 192 ring tile (SEGLAND_CONV_RING192) on the Swin-T stage-2 GEMMs: kernel times on / off, then the Swin-T step A/B/A/B on this box
for s in "32768 768 192" "32768 192 576" "32768 192 192" "32768 192 768" "32768 384 192"; do set -- $s
  for v in 1 0; do echo -n "ring192=$v "; SEGLAND_CONV_RING192=$v python tools/gemm_time.py --tokens $1 --cin $2 --cout $3 2>&1 | grep "^M="; done
done
for rep in 1 2; do for v in 1 0; do
  echo -n "ring192=$v swin_pop: "; SEGLAND_CONV_RING192=$v python bench.py --model swin_pop --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done; done
echo -n "ring192=1 resnet50: "; python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
