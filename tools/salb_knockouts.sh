#!/bin/bash
# Where the split-bf16 saliency forward (csrc/salconvb.hip) spends its time: the same launch with parts switched off (wrong results, timing only).
#   for v in 1 2 6 8 15; do tools/variant_lib.sh salko$v salconvb.hip "-DSALB_KO=$v"; done; tools/salb_knockouts.sh
cd "$(dirname "$0")/.."
run() { CFN_NATIVE_OPS=0 CFN_HIP_LIB=$2 python - <<PY
import sys, torch
sys.path.insert(0, 'coarse-fine-networks_amd')
import cfn_hip
from cfn_hip import ops
cfn_hip.load()
g = torch.Generator().manual_seed(0)
for name, T, H in (('conv1 56->28', 256, 56), ('conv2 28->14', 128, 28)):
    x = torch.randn(8, 24, T, H, H, generator=g).cuda(); w = (torch.randn(24, 24, 3, 3, 3, generator=g) * 0.05).cuda()
    with torch.no_grad():
        for _ in range(3): ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), None, None, 0, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), None, None, 0, True)
        e1.record(); torch.cuda.synchronize()
    print('%-28s %s %.3f ms' % ('$1', name, e0.elapsed_time(e1) / 10))
PY
}
V=coarse-fine-networks_amd/cfn_hip/variants
run "shipped" ""
run "no MFMAs" $V/libcfn_hip_salko1.so
run "no staging arithmetic" $V/libcfn_hip_salko2.so
run "no staging, no loads" $V/libcfn_hip_salko6.so
run "no reduction / stores" $V/libcfn_hip_salko8.so
run "frame loop + barriers only" $V/libcfn_hip_salko15.so
