#!/bin/bash
# Build a variant of libcfn_hip.so with ONE source recompiled under extra flags (same-box A/B of kernel experiments):
#   tools/variant_lib.sh NAME SOURCE.hip "-DFOO=1 ..."   ->  coarse-fine-networks_amd/cfn_hip/variants/libcfn_hip_NAME.so
# (git-ignored, travels with gpurun; select it with CFN_LIB=... in the tools that honour it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/coarse-fine-networks_amd
NAME=$1; SRC=$2; FLAGS=$3
B=$(basename $SRC .hip)
mkdir -p $P/cfn_hip/variants $R/scratch/vobj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result $FLAGS -c $P/csrc/$B.hip -o $R/scratch/vobj/${B}_$NAME.o
OBJS=$(ls $P/build/*.o | grep -v "/$B.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/scratch/vobj/${B}_$NAME.o -o $P/cfn_hip/variants/libcfn_hip_$NAME.so
echo built $P/cfn_hip/variants/libcfn_hip_$NAME.so
