#!/bin/bash
# Round-end evidence, one MI355X box: writes gpurun_out/r3_* (copied into profiles/ afterwards).  usage: bash tools/collect_profiles.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py 2>$O/bench_err.txt | grep '^{"metric"' > $O/r3_bench_default.json
python bench.py --no-cpu-baseline --no-step-graph --profile-table $O/r3_conv_shapes.txt 2>/dev/null | grep '^{"metric"' > $O/r3_bench_kernel_by_kernel.json
python bench.py --backbone resnet101 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $O/r3_bench_r101.json
python bench.py --model swin_pop 2>/dev/null | grep '^{"metric"' > $O/r3_bench_swin.json
for a in "--dtype bf16" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8" "--dtype bf16 --no-step-graph"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/r3_bench_ft.txt
bash tools/ddp_overhead.sh 40 > $O/r3_ddp_overhead.txt 2>/dev/null
bash tools/ddp_overhead.sh 40 --model swin_pop > $O/r3_ddp_overhead_swin.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r50 -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/prof_r50 -name '*kernel_stats.csv' | head -1) $O/r3_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/prof_r50b -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/kernel_sequence.py /tmp/prof_r50b 8 $O/r3_kernel_sequence_graph_step.txt
rocprofv3 --kernel-trace -d /tmp/prof_sw -- python3 $R/bench.py --model swin_pop --steps 6 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/kernel_sequence.py /tmp/prof_sw 8 $O/r3_swin_kernel_sequence_graph_step.txt
rocprofv3 --kernel-trace -d /tmp/prof_sw2 -- python3 $R/bench.py --model swin_pop --steps 5 --warmup 2 --no-cpu-baseline --no-step-graph > /dev/null 2>&1
python3 $R/tools/prof_summary.py /tmp/prof_sw2 7 $O/r3_swin_kernel_stats.txt "bench.py --model swin_pop --steps 5 --warmup 2 --no-step-graph (Swin-T bf16 B=8): rocprofv3 --kernel-trace, all 7 steps" > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-step-graph > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_write -o b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-step-graph > /dev/null 2>&1
cd $R
# the roofline kernel of the bench line (largest total time): the 256 x 256 weight-gradient kernel; then the two conv families the review tracks
python tools/collect_traffic.py /tmp/pmc_fetch /tmp/pmc_write 'conv_wgrad_glds_kernel<unsigned short, 256, 256' 'conv_wgrad_glds_kernel<bf16, 256, 256>' $O/r3_traffic_wgrad.json > /dev/null 2>&1
python tools/collect_traffic.py /tmp/pmc_fetch /tmp/pmc_write 'conv_gemm_p9_kernel' 'conv_gemm_p9_kernel<bf16, 256, 256>' $O/r3_traffic.json > /dev/null 2>&1
python tools/collect_traffic.py /tmp/pmc_fetch /tmp/pmc_write 'conv_gemm_p8_kernel' 'conv_gemm_p8_kernel<bf16, 256, 256>' $O/r3_traffic_p8.json > /dev/null 2>&1
python tools/feed_rate.py --workers 16 2>&1 | tail -1 > $O/r3_feed_rate_now.txt
ls -la $O | grep r3_
head -c 600 $O/r3_bench_default.json; echo
cat $O/r3_ddp_overhead.txt
