#!/bin/bash
# PMC evidence for the two conv kernel families (separate counter passes, --kernel-trace only): MFMA busy share and LDS bank conflicts of the half-tile
# kernel on the 3x3 512 -> 512 d4 layer and of the pixel-stationary kernel on the 256 -> 1024 forward / 1024 -> 256 data gradient.  Output: gpurun_out/r2_pmc_conv.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/r2_pmc_conv.txt
run() {   # name, counters, conv_micro args
  rm -rf /tmp/pmc_$1
  rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmc_$1 -o b -- python3 $R/tools/conv_micro.py $3 > /dev/null 2>&1
  echo "## conv_micro.py $3   --pmc $2" >> $O/r2_pmc_conv.txt
  python3 $R/tools/pmc_summary.py /tmp/pmc_$1 >> $O/r2_pmc_conv.txt 2>&1
}
run p8_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "--cin 512 --cout 512 --k 3 --dil 4 --c1 0 --what fwd,dgrad"
run p8_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "--cin 512 --cout 512 --k 3 --dil 4 --c1 0 --what fwd,dgrad"
run p8_wave "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "--cin 512 --cout 512 --k 3 --dil 4 --c1 0 --what fwd,dgrad"
run sk_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "--cin 256 --cout 1024 --k 1 --c1 0 --what fwd"
run sk_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "--cin 256 --cout 1024 --k 1 --c1 0 --what fwd"
run sk_hbm_f "FETCH_SIZE" "--cin 256 --cout 1024 --k 1 --c1 0 --what fwd"
run sk_hbm_w "WRITE_SIZE" "--cin 256 --cout 1024 --k 1 --c1 0 --what fwd"
cat $O/r2_pmc_conv.txt
