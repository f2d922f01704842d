#!/bin/bash
# Round-end evidence, one MI355X box: writes gpurun_out/r4_* (copied into profiles/ afterwards).  usage: bash tools/collect_profiles.sh
# (counter passes: tools/r4_pmc.sh; feed rates: tools/feed_rate.py; per-shape table of the fine-tune pair: tools/ft_shapes.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py 2>$O/bench_err.txt | grep '^{"metric"' > $O/r4_bench_default.json
python bench.py --no-cpu-baseline --no-other-configs --no-step-graph --profile-table $O/r4_conv_shapes.txt 2>/dev/null | grep '^{"metric"' > $O/r4_bench_kernel_by_kernel.json
python bench.py --backbone resnet101 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $O/r4_bench_r101.json
python bench.py --model swin_pop 2>/dev/null | grep '^{"metric"' > $O/r4_bench_swin.json
for a in "--dtype bf16" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8" "--dtype bf16 --no-step-graph"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/r4_bench_ft.txt
bash tools/ddp_overhead.sh 40 > $O/r4_ddp_overhead.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r50 -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
cp $(find /tmp/prof_r50 -name '*kernel_stats.csv' | head -1) $O/r4_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/prof_r50b -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python3 $R/tools/kernel_sequence.py /tmp/prof_r50b 8 $O/r4_kernel_sequence_graph_step.txt
rocprofv3 --kernel-trace -d /tmp/prof_sw -- python3 $R/bench.py --model swin_pop --steps 6 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/kernel_sequence.py /tmp/prof_sw 8 $O/r4_swin_kernel_sequence_graph_step.txt
rocprofv3 --kernel-trace -d /tmp/prof_ft -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 > /dev/null 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_ft 25 $O/r4_ft_kernel_stats.txt "tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 (one tile pair per step; 5 kernel-by-kernel warm-up steps, capture, 20 replays): rocprofv3 --kernel-trace, averaged over 25 steps" > /dev/null
cd $R
ls -la $O | grep r4_
head -c 700 $O/r4_bench_default.json; echo
