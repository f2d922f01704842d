#!/bin/bash
# Evidence collection on ONE MI355X box; every target writes gpurun_out/<round>_* (copied into profiles/ afterwards).
#   bash tools/collect_profiles.sh [-r r5] <target> [<target> ...]
# targets
#   bench     the default bench line, the kernel-by-kernel line + per-shape conv table, ResNet-101, Swin-T, the fine-tune pair lines
#   stats     rocprofv3 --kernel-trace --stats of the R50 step (kernel by kernel) + the kernel sequence of one graph-replayed step (R50, Swin-T) + fine-tune pair stats
#   pmc       four PMC passes of the R50 step (MFMA busy / clock, LDS + issue, FETCH_SIZE, WRITE_SIZE: separate passes, program directly behind `--`) -> <round>_pmc_families.*, <round>_traffic*.json
#   pmc_swin  the same for the Swin-T step
#   wgrad     tools/wgrad_time.py (nine-tap vs per-tap weight gradients) + tools/wgrad_trace.py phase traces of both kernels
#   ddp       tools/ddp_overhead.sh (DDP / bucket-step host overhead on one GPU)
#   feed      tools/feed_rate.py (tile feed rates)
#   suite     the GPU test suite with durations and the parity log
#   orders    the GPU suite with its files in reverse order (order-dependence screen)
#   nocache   kernel / model tests with torch's caching allocator off (overrun screen)
#   soak      tools/soak.py (drivers for 30 epochs each in one process)
#   odd       bench.py on other batch / tile sizes and modes
set -u
ROUND=r6
if [ "${1:-}" = "-r" ]; then ROUND=$2; shift 2; fi
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
json_line() { grep '^{"metric"'; }
for target in "$@"; do
  cd $R
  case $target in
  bench)
    python bench.py 2>$O/bench_err.txt | json_line > $O/${ROUND}_bench_default.json
    python bench.py --no-cpu-baseline --no-other-configs --no-step-graph --profile-table $O/${ROUND}_conv_shapes.txt 2>/dev/null | json_line > $O/${ROUND}_bench_kernel_by_kernel.json
    python bench.py --backbone resnet101 --no-cpu-baseline --profile-table $O/${ROUND}_conv_shapes_r101.txt 2>/dev/null | json_line > $O/${ROUND}_bench_r101.json
    python bench.py --model swin_pop 2>/dev/null | json_line > $O/${ROUND}_bench_swin.json
    for a in "--dtype bf16" "--dtype bf16 --model swin_pop" "--dtype bf16 --pairs 8" "--dtype bf16 --no-step-graph"; do python tools/bench_ft.py $a 2>/dev/null | grep '^{'; done > $O/${ROUND}_bench_ft.txt
    head -c 900 $O/${ROUND}_bench_default.json; echo ;;
  stats)
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof_r50 /tmp/prof_r50b /tmp/prof_sw /tmp/prof_ft
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r50 -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-step-graph > /dev/null 2>&1
    cp $(find /tmp/prof_r50 -name '*kernel_stats.csv' | head -1) $O/${ROUND}_rocprofv3_kernel_stats.csv
    rocprofv3 --kernel-trace -d /tmp/prof_r50b -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
    python3 $R/tools/kernel_sequence.py /tmp/prof_r50b 8 $O/${ROUND}_kernel_sequence_graph_step.txt
    rocprofv3 --kernel-trace -d /tmp/prof_sw -- python3 $R/bench.py --model swin_pop --steps 6 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
    python3 $R/tools/kernel_sequence.py /tmp/prof_sw 8 $O/${ROUND}_swin_kernel_sequence_graph_step.txt
    rocprofv3 --kernel-trace -d /tmp/prof_ft -- python3 $R/tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 > /dev/null 2>&1
    python3 $R/tools/prof_summary.py /tmp/prof_ft 25 $O/${ROUND}_ft_kernel_stats.txt "tools/bench_ft.py --dtype bf16 --steps 20 --warmup 5 (one tile pair per step; 5 kernel-by-kernel warm-up steps, capture, 20 replays): rocprofv3 --kernel-trace, averaged over 25 steps" > /dev/null ;;
  pmc|pmc_swin)
    cd /tmp && export TMPDIR=/tmp
    if [ $target = pmc ]; then ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-step-graph --no-other-configs"; TAG=""; else ARGS="--model swin_pop --steps 3 --warmup 1 --no-cpu-baseline --no-step-graph"; TAG="_swin"; fi
    rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC /tmp/pmcD
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcA -o b -- python3 $R/bench.py $ARGS > $O/pmc_passA.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmcB -o b -- python3 $R/bench.py $ARGS > $O/pmc_passB.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcC -o b -- python3 $R/bench.py $ARGS > $O/pmc_passC.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcD -o b -- python3 $R/bench.py $ARGS > $O/pmc_passD.log 2>&1
    cd $R
    python3 tools/pmc_families.py --cmd "python3 bench.py $ARGS" $O/${ROUND}_pmc_families$TAG.txt $O/${ROUND}_pmc_families$TAG.json 7 /tmp/pmcA /tmp/pmcB /tmp/pmcC /tmp/pmcD > $O/pmc_summary.log 2>&1
    if [ $target = pmc ]; then
      for k in "conv_gemm_p9_kernel:conv_gemm_p9_kernel<bf16, 256, 256>:traffic.json" "conv_gemm_p8_kernel:conv_gemm_p8_kernel<bf16, 256, 256>:traffic_p8.json" "conv_wgrad3_kernel:conv_wgrad3_kernel:traffic_wgrad.json" "bn_bwd_reduce_kernel:bn_bwd_reduce_kernel:traffic_bn_bwd_reduce.json"; do
        IFS=: read needle label file <<< "$k"
        python3 tools/collect_traffic.py /tmp/pmcC /tmp/pmcD "$needle" "$label" $O/${ROUND}_$file > /dev/null 2>&1
      done
      # a `bench` target behind this one in the same call reads the counters from profiles/ (sha-gated on the library): give it this box's fresh ones
      cp $O/${ROUND}_pmc_families.json $O/${ROUND}_traffic*.json $R/profiles/ 2>/dev/null
    fi
    head -30 $O/${ROUND}_pmc_families$TAG.txt ;;
  wgrad)
    python tools/wgrad_time.py 2>&1 | grep -v amdgpu.ids > $O/${ROUND}_wgrad3_time.txt
    : > $O/${ROUND}_wgrad_trace.txt
    for s in "256 256 2" "512 512 4" "2048 512 1" "128 128 1"; do set -- $s; python tools/wgrad_trace.py --cin $1 --cout $2 --dil $3 --both 2>&1 | grep -v amdgpu.ids >> $O/${ROUND}_wgrad_trace.txt; done
    cat $O/${ROUND}_wgrad3_time.txt ;;
  ddp)     bash tools/ddp_overhead.sh 40 > $O/${ROUND}_ddp_overhead.txt 2>/dev/null ;;
  feed)    python tools/feed_rate.py > $O/${ROUND}_feed_rate.txt 2>&1 ;;
  suite)
    rm -f $O/${ROUND}_parity_log.txt
    SEGLAND_PARITY_LOG=$O/${ROUND}_parity_log.txt python -m pytest tests -m gpu -q --durations=25 2>&1 | grep -v Warning | tail -45 > $O/${ROUND}_pytest_durations.txt
    tail -3 $O/${ROUND}_pytest_durations.txt ;;
  orders)    # the driver runs the files alphabetically; round 4's workspace fault only showed in another order
    REV=$(ls tests/test_*gpu*.py | sort -r | tr '\n' ' ')
    timeout 1500 python -m pytest $REV -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v Warning | tail -3 > $O/${ROUND}_pytest_other_orders.txt
    SHUF="tests/test_round3_gpu.py tests/test_swin_gpu.py tests/test_round5_gpu.py tests/test_drivers_gpu.py tests/test_round4_gpu.py tests/test_kernels_gpu.py tests/test_graph_step_gpu.py tests/test_round2_gpu.py tests/test_model_gpu.py"
    timeout 1500 python -m pytest $SHUF -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v Warning | tail -3 >> $O/${ROUND}_pytest_other_orders.txt
    cat $O/${ROUND}_pytest_other_orders.txt ;;
  nocache)   # every tensor its own hipMalloc: an access past the end of a buffer meets an unmapped page far more often; graphs off (a capture needs the caching allocator)
    : > $O/${ROUND}_nocache_screen.txt
    for f in test_kernels_gpu test_model_gpu test_swin_gpu test_round2_gpu test_round3_gpu test_round4_gpu test_round5_gpu; do
      PYTORCH_NO_CUDA_MEMORY_CACHING=1 SEGLAND_STEP_GRAPH=0 SEGLAND_FEATURE_GRAPH=0 timeout 1500 python -X faulthandler -m pytest tests/$f.py -m gpu -q -x -p no:cacheprovider \
        -k "not graph and not drivers and not two_ranks and not rccl and not bucket and not capture and not feature and not workers" 2>&1 | grep -v Warning | tail -1 | sed "s/^/$f: /" >> $O/${ROUND}_nocache_screen.txt
    done
    cat $O/${ROUND}_nocache_screen.txt ;;
  soak)    timeout 1500 python tools/soak.py --epochs 30 > $O/${ROUND}_soak.txt 2>&1; echo "rc $?" >> $O/${ROUND}_soak.txt; tail -5 $O/${ROUND}_soak.txt ;;
  odd)     # bench.py on configurations other than the headline one: must run and print a line
    : > $O/${ROUND}_odd_shapes.txt
    for a in "--batch 3 --size 256" "--batch 2 --size 640" "--batch 5 --size 384 --dtype f32" "--model swin_pop --batch 3 --size 448" "--backbone resnet101 --batch 7 --size 320"; do
      echo "== bench.py $a" >> $O/${ROUND}_odd_shapes.txt
      timeout 600 python bench.py $a --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs 2>$O/odd_err.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['config'])" >> $O/${ROUND}_odd_shapes.txt 2>&1 || tail -5 $O/odd_err.txt >> $O/${ROUND}_odd_shapes.txt
    done
    cat $O/${ROUND}_odd_shapes.txt ;;
  *) echo "unknown target $target"; exit 2 ;;
  esac
done
ls -la $O | grep " ${ROUND}_" | tail -40
