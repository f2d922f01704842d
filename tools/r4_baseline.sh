#!/bin/bash
# round-4 baseline: GPU suite durations + per-kernel tables of the ft pair step and the swin step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=60 > $O/pytest_durations.txt 2>&1
python tools/bench_ft.py --dtype bf16 2>/dev/null | grep '^{' > $O/ft.txt
python tools/bench_ft.py --dtype bf16 --no-step-graph 2>/dev/null | grep '^{' >> $O/ft.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_ft -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 10 --warmup 3 --no-step-graph > /dev/null 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_ft 13 $O/ft_kernel_stats.txt "bench_ft --dtype bf16 --no-step-graph, 13 steps" > /dev/null
rocprofv3 --kernel-trace -d /tmp/prof_ft2 -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 10 --warmup 5 > /dev/null 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_ft2 15 $O/ft_kernel_stats_graph.txt "bench_ft --dtype bf16 (graph), 15 steps" > /dev/null
cd $R; python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $O/bench.json
