#!/usr/bin/env python3
"""Runs only the pointwise contractions of X3D-M layers 2-4 (forward, data gradient, weight gradient; 8 clips x T frames by
default) -- the workload of the rocprofv3 --pmc passes behind profiles/r03_pmc_pw_mfma.json.
env: T, B, REPS, CFN_PW_SPLIT (0 = fp32 MFMA, 3 / 6 = split-bf16 terms), DT (f32 | bf16 activations), ONLY (substring of a layer name)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
from cfn_hip import ops
T = int(os.environ.get('T', '256'))
NB = int(os.environ.get('B', '8'))
REPS = int(os.environ.get('REPS', '3'))
DT = torch.bfloat16 if os.environ.get('DT', 'f32') == 'bf16' else torch.float32
ONLY = os.environ.get('ONLY')
LAYERS = [('L2 conv3 108->48 @28', 108, 48, 28), ('L2 conv1 48->108 @28', 48, 108, 28), ('L3 conv1 96->216 @14', 96, 216, 14),
          ('L3 conv3 216->96 @14', 216, 96, 14), ('L4 conv1 192->432 @7', 192, 432, 7), ('L4 conv3 432->192 @7', 432, 192, 7)]
for name, ci, co, H in LAYERS:
    if ONLY and ONLY not in name:
        continue
    x = torch.randn(NB, ci, T, H, H, device='cuda').to(DT).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, 1, device='cuda') * 0.1).requires_grad_(True)
    A = (torch.rand(NB, ci, device='cuda') + 0.5).requires_grad_(True)
    B = (torch.randn(NB, ci, device='cuda') * 0.1).requires_grad_(True)
    for _ in range(REPS + 1):
        y, sm, sq = ops.pwconv(x, w, A, B, 2, 1, True)
        gy, gs, gq = torch.randn_like(y), torch.randn_like(sm) * 0.01, torch.randn_like(sq) * 0.001
        torch.autograd.grad((y, sm, sq), (x, w, A, B), (gy, gs, gq))
    del x, y, gy
torch.cuda.synchronize()
print('done')
