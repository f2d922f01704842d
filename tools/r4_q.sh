#!/bin/bash
# round 4: row-parallel pyramid cell sums for small batches + 64-row ring tiles: tests, fine-tune benches, kernel statistics
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_model_gpu.py tests/test_round2_gpu.py -k "not two_ranks and not rccl" -m gpu -q -x > $O/pytest_subset.txt 2>&1; echo "rc $?" >> $O/pytest_subset.txt
for a in "--dtype bf16" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/bench_ft.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_ft -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 > /dev/null 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_ft 25 $O/ft_kernel_stats.txt "tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 (one tile pair per step; 5 kernel-by-kernel warm-up steps, capture, 20 replays)" > /dev/null
