#!/bin/bash
# Per-kernel register / spill / LDS / occupancy table of every HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
#   tools/kernel_resources.sh out.txt      -- to compare two states of the tree:  diff <(sort a.txt) <(sort b.txt)
out=${1:-/dev/stdout}
cd "$(dirname "$0")/../coarse-fine-networks_amd/csrc"
tmp=$(mktemp -d)
ls *.hip | xargs -P 8 -I{} sh -c "extra=\$(grep -h '^// hipcc-flags:' {} | cut -d: -f2-); /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result \$extra -c {} -o /dev/null --cuda-device-only -Rpass-analysis=kernel-resource-usage 2> $tmp/{}.log"
for f in $tmp/*.log; do
  grep "Function Name\|    VGPRs:\|AGPRs\|ScratchSize\|Occupancy\|VGPRs Spill\|LDS Size" "$f" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - - | sed "s|^|$(basename $f .hip.log): |"
done | sed 's/Function Name: //; s/\t */ | /g' > "$out"
rm -rf $tmp
