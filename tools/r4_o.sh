#!/bin/bash
# round 4: timings of the multi-process tests with the children bound to 16 OpenMP threads; bench lines of the three models on the same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -k "two_ranks or rccl" -m gpu -q -x --durations=10 > $O/pytest_children.txt 2>&1; echo "rc $?" >> $O/pytest_children.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --model swin_pop --backbone swin-t --batch 8 --no-cpu-baseline --no-other-configs > $O/bench_swin.json 2> $O/bench_swin.err
python bench.py --backbone resnet101 --no-cpu-baseline --no-other-configs > $O/bench_r101.json 2> $O/bench_r101.err
for a in "--dtype bf16" "--dtype bf16 --torch-sgd" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/bench_ft.txt
