#!/bin/bash
# Ablation builds of conv_wgrad_glds_kernel (round 6): the same library with the loop's LDS-DMA issue (bit 0) and / or its fragment reads (bit 1) compiled out, to
# see what each costs per 32-row stage (results are garbage; only the phase trace matters).  Here:  bash tools/wgrad_ablation.sh build   (three extra .so, git-ignored)
# On the GPU box:  bash tools/wgrad_ablation.sh run > gpurun_out/r6_wgrad_ablation.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/segland_amd/csrc
if [ "$1" = build ]; then
  for v in 1 2 3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSL_WG_ABL=$v -c $C/conv_wgrad.hip -o $C/conv_wgrad_abl$v.o || exit 1
    objs=$(ls $C/*.o | grep -v conv_wgrad.o | grep -v conv_wgrad_abl)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $C/conv_wgrad_abl$v.o -o $C/libsegland_abl$v.so || exit 1
    rm -f $C/conv_wgrad_abl$v.o
    echo built libsegland_abl$v.so
  done
elif [ "$1" = run ]; then
  for shape in "512 2048" "256 1024" "1024 256"; do set -- $shape
    for v in 0 1 2 3; do
      case $v in 0) lib=$C/libsegland_hip.so; what="product";; 1) lib=$C/libsegland_abl1.so; what="no LDS-DMA in the loop";; 2) lib=$C/libsegland_abl2.so; what="no fragment reads in the loop";; 3) lib=$C/libsegland_abl3.so; what="MFMAs + barriers only";; esac
      echo "#### $what"
      SEGLAND_LIB_PATH=$lib python $R/tools/wgrad_trace.py --k 1 --dil 1 --cin $1 --cout $2 2>&1 | grep -v amdgpu.ids | head -3
    done
  done
fi

# ring-depth A/B builds:  bash tools/wgrad_ablation.sh build_nst   ->  libsegland_nst5.so (256 x 256: 5 stages; 128 x 256: 5), libsegland_nst6.so (5 / 6)
if [ "$1" = build_nst ]; then
  for v in "5 5 nst5" "5 6 nst6"; do set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSL_WG_NST_256=$1 -DSL_WG_NST_384=$2 -c $C/conv_wgrad.hip -o $C/conv_wgrad_$3.o || exit 1
    objs=$(ls $C/*.o | grep -v conv_wgrad.o | grep -v conv_wgrad_nst)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $C/conv_wgrad_$3.o -o $C/libsegland_$3.so || exit 1
    rm -f $C/conv_wgrad_$3.o
    echo built libsegland_$3.so
  done
fi
if [ "$1" = run_nst ]; then
  for shape in "512 2048" "256 1024" "1024 256" "1024 2048"; do set -- $shape
    for v in hip nst5 nst6; do
      echo "#### lib $v"
      SEGLAND_LIB_PATH=$C/libsegland_$v.so python $R/tools/wgrad_trace.py --k 1 --dil 1 --cin $1 --cout $2 2>&1 | grep -v amdgpu.ids | head -3
    done
  done
  for v in hip nst5 nst6 hip nst5 nst6; do echo -n "lib $v: "; SEGLAND_LIB_PATH=$C/libsegland_$v.so python $R/bench.py --no-cpu-baseline --no-other-configs --steps 100 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['families']; print(d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in f.items() if 'wgrad_glds' in k})"; done
fi
