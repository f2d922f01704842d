#!/bin/bash
# Ablation builds of conv_gemm_sk_kernel (round 6): what the pixel-stationary kernel's MODE 5 store loop (layer3 conv1 data gradient + shortcut + gate + bn3 column sums:
# 116 us at 3.9 TB/s, the largest single launch class of a ResNet-101 step) spends its time on.  -DSL_SK_ABL bits: 1 no global stores, 2 no addend / BN-input loads,
# 4 no column-sum arithmetic, 8 no MFMAs (results are garbage; only the time matters).   here: bash tools/sk_ablation.sh build    GPU box: bash tools/sk_ablation.sh run
set -u
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/segland_amd/csrc
if [ "$1" = build ]; then
  for v in 1 2 3 4 8 15; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSL_SK_ABL=$v -c $C/conv_gemm_sk.hip -o /tmp/sk_abl$v.o || exit 1
    objs=$(ls $C/*.o | grep -v conv_gemm_sk.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sk_abl$v.o -o $C/libsegland_skabl$v.so || exit 1
    echo built libsegland_skabl$v.so
  done
elif [ "$1" = run ]; then
  for v in 0 1 2 3 4 8 15; do
    case $v in 0) lib=libsegland_hip.so; what="product";; 1) what="no global stores";; 2) what="no addend / BN-input loads";; 3) what="no stores, no loads";; 4) what="no column-sum arithmetic";; 8) what="no MFMAs";; 15) what="ring + barriers only";; esac
    [ $v != 0 ] && lib=libsegland_skabl$v.so
    echo "#### $what"
    SEGLAND_LIB_PATH=$C/$lib python $R/tools/sk_mode5_time.py 2>&1 | grep -v amdgpu.ids
  done
fi
