#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q -x --durations=10 > $O/pytest_r4.txt 2>&1; echo "rc $?" >> $O/pytest_r4.txt
python tools/ft_shapes.py > $O/ft_shapes.txt 2>&1
python tools/ft_shapes.py --tiles256 1000 > $O/ft_shapes_no256.txt 2>&1
for a in "--dtype bf16" "--dtype bf16 --no-step-graph" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/ft.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_graph_step_gpu.py tests/test_swin_gpu.py -m gpu -q -x --durations=10 > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt
python bench.py --no-cpu-baseline 2>$O/bench_err.txt | grep '^{"metric"' > $O/bench.json
python bench.py --no-cpu-baseline --model swin_pop 2>/dev/null | grep '^{"metric"' > $O/bench_swin.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4c -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/prof_r4c -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/prof_ft3 -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 10 --warmup 5 > /dev/null 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_ft3 15 $O/ft_kernel_stats_graph.txt "bench_ft --dtype bf16 (graph), 15 steps" > /dev/null
