#!/bin/bash
# round-4 counter evidence on the final build: four PMC passes of the bench, program directly after `--`
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-step-graph --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcA -o b -- python3 $R/bench.py $ARGS > $O/passA.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmcB -o b -- python3 $R/bench.py $ARGS > $O/passB.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcC -o b -- python3 $R/bench.py $ARGS > $O/passC.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcD -o b -- python3 $R/bench.py $ARGS > $O/passD.log 2>&1
cd $R
python3 tools/pmc_families.py $O/r4_pmc_families.txt $O/r4_pmc_families.json 7 /tmp/pmcA /tmp/pmcB /tmp/pmcC /tmp/pmcD > $O/summary.log 2>&1
for k in "conv_gemm_p9_kernel:conv_gemm_p9_kernel<bf16, 256, 256>:r4_traffic.json" "conv_gemm_p8_kernel:conv_gemm_p8_kernel<bf16, 256, 256>:r4_traffic_p8.json" "conv_wgrad_glds_kernel<unsigned short, 256, 256:conv_wgrad_glds_kernel<bf16, 256, 256>:r4_traffic_wgrad.json" "bn_bwd_reduce_kernel:bn_bwd_reduce_kernel:r4_traffic_bn_bwd_reduce.json"; do
  IFS=: read needle label file <<< "$k"
  python3 tools/collect_traffic.py /tmp/pmcC /tmp/pmcD "$needle" "$label" $O/$file > /dev/null 2>&1
done
ls -la $O
