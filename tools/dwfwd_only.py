#!/usr/bin/env python3
"""Runs only the depthwise-conv forward stack of x3d_fine X3D-M (conv1_t + 26 conv2 launches) at T frames, batch B
(env T, B) -- the workload of the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes behind bench.py's roofline.traffic.
Each distinct layer shape runs once per occurrence in the network, after one untimed warm-up launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
from cfn_hip import ops
T = int(os.environ.get('T', '256'))
NB = int(os.environ.get('B', '8'))
x = torch.randn(NB, 24, T, 112, 112, device='cuda')
w = torch.randn(24, 1, 5, 1, 1, device='cuda') * 0.3
for _ in range(2):
    ops.dwconv_t5(x, w, True)
del x
# (channels, H_in, stride, occurrences in X3D-M)
LAYERS = [(54, 112, 2, 1), (54, 56, 1, 2), (108, 56, 2, 1), (108, 28, 1, 4), (216, 28, 2, 1), (216, 14, 1, 10), (432, 14, 2, 1), (432, 7, 1, 6)]
for c, H, s, reps in LAYERS:
    x = torch.randn(NB, c, T, H, H, device='cuda')
    w = torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2
    A = torch.rand(NB, c, device='cuda') + 0.5
    B = torch.randn(NB, c, device='cuda') * 0.1
    for _ in range(reps + 1):
        ops.dwconv3d(x, w, A, B, 1, s, True)
    del x
torch.cuda.synchronize()
print('done')
