#!/usr/bin/env python3
"""Sustained timing of the fused depthwise backward entry point (cfn_dwconv3d_bwd_fused) on one X3D-M conv2 shape, 8 clips x T = 256:
    python tools/dwbwd_busy.py 7 [--secs 3]      # plane size: 56 | 28 | 14 | 7"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import cfn_hip
ap = argparse.ArgumentParser()
ap.add_argument('plane', type=int)
ap.add_argument('--secs', type=float, default=3.0)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--frames', type=int, default=256)
ap.add_argument('--stride', type=int, default=1, help='2: the plane is the INPUT plane (112 | 56 | 28 | 14), cfn_dwconv3d_bwd_fused_s2')
a = ap.parse_args()
cfn_hip.load()
H = a.plane
C = {56: 54, 28: 108, 14: 216, 7: 432}[H] if a.stride == 1 else {112: 54, 56: 108, 28: 216, 14: 432}[H]
Ho = H // a.stride
B, T = a.batch, a.frames
dev = 'cuda'
gy, y = (torch.randn(B, C, T, Ho, Ho, device=dev) for _ in range(2))
x = torch.randn(B, C, T, H, H, device=dev)
gx = torch.empty_like(x)
w = torch.randn(C, 27, device=dev) * 0.2
gs, gq = (torch.randn(B, C, device=dev, dtype=torch.float64) * 0.01 for _ in range(2))
A = (torch.rand(B, C, device=dev) + 0.5).double()
Bc = (torch.randn(B, C, device=dev) * 0.1).double()
gA, gB = (torch.zeros(B, C, dtype=torch.float64, device=dev) for _ in range(2))
gw = torch.zeros(C, 27, dtype=torch.float64, device=dev)
fn = lambda: cfn_hip.call('cfn_dwconv3d_bwd_fused' + ('_s2' if a.stride == 2 else ''), gy, y, gs, gq, w, x, A, Bc, 1, gx, gA, gB, gw, B, C, T, H, H)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
t0 = time.time()
while time.time() - t0 < a.secs:
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 20)
gb = 4.0 * B * C * T * (2.0 * H * H + 2.0 * Ho * Ho) / 1e9
print(('dw bwd %dx%d' + (' stride 2' if a.stride == 2 else '') + ': first %.1f us, min %.1f us = %.2f TB/s (gy, y, x read, gx written)') % (H, H, ms[0] * 1e3, min(ms) * 1e3, gb / min(ms)))
