#!/usr/bin/env python3
"""Backward time of the 14 -> 7 depthwise conv (432 channels, 8 clips x T = 256) through the autograd wrapper (events around backward()):
    python tools/dwbwd_s2_time.py;  CFN_DW_FLATB=8 python tools/dwbwd_s2_time.py   # flat kernel off: dgrad + wgrad band kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
from cfn_hip import ops
B, T, C, H = 8, 256, 432, 14
x = torch.randn(B, C, T, H, H, device='cuda', requires_grad=True)
w = (torch.randn(C, 1, 3, 3, 3, device='cuda') * 0.2).requires_grad_(True)
A = (torch.rand(B, C, device='cuda') + 0.5).requires_grad_(True)
Bc = (torch.randn(B, C, device='cuda') * 0.1).requires_grad_(True)
best = 1e9
for it in range(30):
    y, s, q = ops.dwconv3d(x, w, A, Bc, 1, 2, True)
    loss = (y * 0.5).sum() + s.sum() * 0.01 + q.sum() * 0.001
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loss.backward()
    e1.record(); torch.cuda.synchronize()
    if it >= 5:
        best = min(best, e0.elapsed_time(e1))
    x.grad = w.grad = A.grad = Bc.grad = None
print('14->7 backward (incl. the sum / scale glue of the loss): %.1f us' % (best * 1e3))
