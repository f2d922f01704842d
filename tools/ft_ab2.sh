#!/bin/bash
# fine-tune pair, same box: the frozen base chain kept across iterations (SEGLAND_BASE_CHAIN_CACHE) and the clip coefficient inside the SGD launch (SEGLAND_FT_CLIP_IN_STEP)
run() { python tools/bench_ft.py --dtype bf16 $1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'])"; }
for rep in 1 2; do
  echo -n "both on:                    "; run
  echo -n "SEGLAND_BASE_CHAIN_CACHE=0: "; SEGLAND_BASE_CHAIN_CACHE=0 run
  echo -n "SEGLAND_FT_CLIP_IN_STEP=0:  "; SEGLAND_FT_CLIP_IN_STEP=0 run
  echo -n "both off:                   "; SEGLAND_BASE_CHAIN_CACHE=0 SEGLAND_FT_CLIP_IN_STEP=0 run
done
echo -n "swin, both on:  "; run "--model swin_pop"
echo -n "swin, both off: "; SEGLAND_BASE_CHAIN_CACHE=0 SEGLAND_FT_CLIP_IN_STEP=0 run "--model swin_pop"
