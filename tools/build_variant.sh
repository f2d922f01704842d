#!/bin/bash
# A variant build of libsegland_hip.so for a same-box A/B (tools/ab_libs.sh): ONE source file recompiled with extra -D flags, linked with the product's other objects.
#   bash tools/build_variant.sh <name> <source.hip> -DSL_RING_LATE=0 [...]   ->  segland_amd/csrc/libsegland_<name>.so (git-ignored; delete after the A/B)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/segland_amd/csrc
NAME=$1; SRC=$2; shift 2
make -C $C -j8 > /dev/null
O=$C/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $C/$SRC -o $O
objs=$(ls $C/*.o | grep -v "/${SRC%.hip}.o" | grep -v "_$NAME.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $O -o $C/libsegland_$NAME.so
rm -f $O
echo built libsegland_$NAME.so
