#!/bin/bash
# round 4: the counter passes of tools/r4_pmc.sh for the Swin-T step (config 5) -- what binds a step whose largest kernel family has 12 % of the time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4pmc_swin; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--model swin_pop --steps 3 --warmup 1 --no-cpu-baseline --no-step-graph --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcA -o b -- python3 $R/bench.py $ARGS > $O/passA.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmcB -o b -- python3 $R/bench.py $ARGS > $O/passB.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcC -o b -- python3 $R/bench.py $ARGS > $O/passC.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcD -o b -- python3 $R/bench.py $ARGS > $O/passD.log 2>&1
cd $R
python3 tools/pmc_families.py --min-ms 0.09 --cmd "python3 bench.py $ARGS" $O/r4_pmc_families_swin.txt $O/r4_pmc_families_swin.json 7 /tmp/pmcA /tmp/pmcB /tmp/pmcC /tmp/pmcD > $O/summary.log 2>&1
ls -la $O
