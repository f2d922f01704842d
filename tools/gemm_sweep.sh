#!/bin/bash
# the Swin-T GEMM shapes of one step (B = 8, 512 x 512) under a set of environment switches.  usage: bash tools/gemm_sweep.sh "ENV=.." ...
for envs in "" "$@"; do
  echo "== ${envs:-default}"
  for cfg in "--tokens 131072 --cin 128 --cout 384" "--tokens 131072 --cin 128 --cout 128 --res" "--tokens 131072 --cin 128 --cout 384 --gelu" "--tokens 131072 --cin 384 --cout 128 --res" \
             "--tokens 32768 --cin 192 --cout 576" "--tokens 32768 --cin 192 --cout 192 --res" "--tokens 32768 --cin 192 --cout 768 --gelu" "--tokens 32768 --cin 768 --cout 192 --res" \
             "--tokens 8192 --cin 384 --cout 1152" "--tokens 8192 --cin 384 --cout 384 --res" "--tokens 8192 --cin 384 --cout 1536 --gelu" "--tokens 8192 --cin 1536 --cout 384 --res" \
             "--tokens 2048 --cin 768 --cout 2304" "--tokens 2048 --cin 768 --cout 768 --res" "--tokens 2048 --cin 768 --cout 3072 --gelu" "--tokens 2048 --cin 3072 --cout 768 --res"; do
    env $envs python3 tools/gemm_time.py $cfg 2>/dev/null | grep "^M="
  done
done
