#!/usr/bin/env python3
"""Keeps the GPU busy with ONE kernel for --secs seconds (for tools/clk_watch.sh): copy | t5 | dw<H>s<stride> | pw"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
from cfn_hip import ops
ap = argparse.ArgumentParser()
ap.add_argument('what')
ap.add_argument('--secs', type=float, default=6.0)
a = ap.parse_args()
B, T = 8, 256
if a.what == 'copy':
    x = torch.randn(B, 54, T, 56, 56, device='cuda'); y = torch.empty_like(x)
    fn = lambda: y.copy_(x)
    gb = 2 * x.numel() * 4 / 1e9
elif a.what == 't5':
    x = torch.randn(B, 24, T, 112, 112, device='cuda'); w = torch.randn(24, 1, 5, 1, 1, device='cuda') * 0.3
    fn = lambda: ops.dwconv_t5(x, w, True)
    gb = 2 * x.numel() * 4 / 1e9
elif a.what.startswith('dw'):
    H, s = a.what[2:].split('s'); H = int(H); s = int(s)
    c = {112: 54, 56: 54 if s == 1 else 108, 28: 108 if s == 1 else 216, 14: 216 if s == 1 else 432, 7: 432}[H]
    x = torch.randn(B, c, T, H, H, device='cuda'); w = torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2
    A = torch.rand(B, c, device='cuda') + 0.5; Bb = torch.randn(B, c, device='cuda') * 0.1
    fn = lambda: ops.dwconv3d(x, w, A, Bb, 1, s, True)
    gb = x.numel() * 4 * (1 + 1.0 / (s * s)) / 1e9
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
while time.time() - t0 < a.secs:
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 50)
print('%s: first %.1f us, last %.1f us, min %.1f us = %.2f TB/s' % (a.what, ms[0] * 1e3, ms[-1] * 1e3, min(ms) * 1e3, gb / min(ms)))
