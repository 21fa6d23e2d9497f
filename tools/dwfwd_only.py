#!/usr/bin/env python3
"""Runs only the depthwise-conv forward stack of X3D-M at T=256 (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
from cfn_hip import ops
T = int(os.environ.get('T', '256'))
LAYERS = [(54, 112, 2, 1), (54, 56, 1, 2), (108, 56, 2, 1), (108, 28, 1, 4), (216, 28, 2, 1), (216, 14, 1, 10), (432, 14, 2, 1), (432, 7, 1, 6)]
for c, H, s, reps in LAYERS:
    x = torch.randn(1, c, T, H, H, device='cuda')
    w = torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2
    A = torch.rand(1, c, device='cuda') + 0.5
    B = torch.randn(1, c, device='cuda') * 0.1
    for _ in range(reps + 1):
        ops.dwconv3d(x, w, A, B, 1, s, True)
torch.cuda.synchronize()
print('done')
