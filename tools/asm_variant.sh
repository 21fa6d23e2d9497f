#!/bin/bash
# Build a variant of libcfn_hip.so in which the DEVICE code of ONE source file comes from a hand-edited assembly file
# (ISA-level bisection: hipcc -S once, edit the .s, nothing else in the library moves).
#   tools/asm_variant.sh <tree> <source.hip> <edited.s> <out.so>
# <tree> is a checkout with a finished build (coarse-fine-networks_amd/build/*.o); the host side of <source.hip> is
# compiled as usual, its fat binary is replaced by the assembled <edited.s>, and the library is relinked.
set -e
TREE=$(realpath "$1"); SRC=$2; ASM=$(realpath "$3"); OUT=$(realpath -m "$4")
LLVM=/opt/rocm/lib/llvm/bin
CSRC=$TREE/coarse-fine-networks_amd/csrc
OBJ=$TREE/coarse-fine-networks_amd/build
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$ASM" -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $W/dev.o -o $W/dev.out
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result \
    --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $CSRC/$SRC -o $W/host.o
base=$(basename $SRC .hip)
objs=$(ls $OBJ/*.o | grep -v "/$base.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $W/host.o -o "$OUT"
echo "built $OUT"
