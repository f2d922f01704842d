#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_round2_gpu.py -k "g1 or g2 or g7 or head or ft or decompose or argmax" -m gpu -q -x --durations=5 > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt
for a in "--dtype bf16" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/ft.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_ft -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 > /dev/null 2>&1
python3 $R/tools/step_gaps.py /tmp/prof_ft stem_conv_fwd 8 > $O/ft_gaps.txt 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_ft 25 $O/ft_kernel_stats.txt "bench_ft 25 steps" > /dev/null
