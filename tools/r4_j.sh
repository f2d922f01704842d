#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
for pm in 0 16384; do echo "== pair_min $pm"; for cfg in "--tokens 32768 --cin 192 --cout 576" "--tokens 32768 --cin 192 --cout 192" "--tokens 32768 --cin 192 --cout 768" "--tokens 32768 --cin 768 --cout 192" "--tokens 131072 --cin 128 --cout 384" "--tokens 8192 --cin 384 --cout 1152"; do python tools/gemm_time.py $cfg --pair-min $pm 2>&1 | grep "^M="; done; done > $O/gemm_pair.txt
for pm in 524288 16384; do echo "== pair_min $pm"; python tools/with_hook.py sl_debug_wgrad_pair_min=$pm -- bench.py --model swin_pop --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done > $O/swin_pair.txt
