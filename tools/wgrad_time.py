#!/usr/bin/env python3
"""Sustained time of cfn_pwconv_bwd_weight on one shape: python tools/wgrad_time.py Cin Cout H [T] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import cfn_hip
cfn_hip.load()
K, M, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
N = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = 'cuda'
gy, y = (torch.randn(N, M, T, H, H, device=dev) for _ in range(2))
x = torch.randn(N, K, T, H, H, device=dev)
gs, gq = (torch.randn(N, M, device=dev).double() * 0.01 for _ in range(2))
A, B = (torch.rand(N, K, device=dev) + 0.5).double(), (torch.randn(N, K, device=dev) * 0.1).double()
gw = torch.zeros(M, K, dtype=torch.float64, device=dev)
fn = lambda: cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, 1, gw, N, K, M, T, H, H, 1, None)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
t0 = time.time()
while time.time() - t0 < 2.0:
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
gb = 4.0 * N * T * H * H * (2 * M + K) / 1e9
print('wgrad %d -> %d @%dx%d T=%d: %.1f us = %.2f TB/s' % (K, M, H, H, T, best * 1e3, gb / best))
