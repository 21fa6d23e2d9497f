#!/usr/bin/env python3
"""Steady-state timing of the stem weight gradient (cfn_stem_conv_bwd_weight) at 8 x 3 x 256 x 224 x 224:
    python tools/stem_wg_bench.py            # stem_wgrad_kernel
    CFN_STEM_WG_OFF=1 python tools/stem_wg_bench.py   # implicit-GEMM path (pw_wgrad_direct_kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import cfn_hip
cfn_hip.load()
B, T = int(os.environ.get('B', 8)), int(os.environ.get('T', 256))
x = torch.randn(B, 3, T, 224, 224, device='cuda')
gy = torch.randn(B, 24, T, 112, 112, device='cuda')
gw = torch.zeros(24, 27, dtype=torch.float64, device='cuda')
fn = lambda: cfn_hip.call('cfn_stem_conv_bwd_weight', gy, x, gw, B, 3, 24, T, 224, 224)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
t0 = time.time()
while time.time() - t0 < 3.0:
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 20)
gb = 4.0 * B * T * (3 * 224 * 224 + 24 * 112 * 112) / 1e9
print('stem wgrad: first %.1f us, min %.1f us = %.2f TB/s' % (ms[0] * 1e3, min(ms) * 1e3, gb / min(ms)))
