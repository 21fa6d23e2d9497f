#!/usr/bin/env python3
"""Per-launch time of bn_fold forward + backward (autograd wrapper, 8 samples, 8 split groups) with and without the squeeze-excite gate on the
X3D-M conv2 widths -- back-to-back launches between two events (includes the wrapper's allocations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
from cfn_hip import ops
dev = 'cuda'
N, S = 8, 8
for C, Wd in ((54, 8), (108, 8), (216, 16), (432, 32)):
    for se in (False, True):
        s = torch.randn(N, C, device=dev, dtype=torch.float64).abs() * 1e5
        q = s * s / 1e5 + 1e6
        s.requires_grad_(True); q.requires_grad_(True)
        gamma = torch.ones(C, device=dev, requires_grad=True); beta = torch.zeros(C, device=dev, requires_grad=True)
        bufs = (torch.zeros(S * C, device=dev), torch.ones(S * C, device=dev), torch.zeros(1, dtype=torch.long, device=dev))
        sew = None
        if se:
            sew = tuple(t.requires_grad_(True) for t in (torch.randn(Wd, C, 1, 1, 1, device=dev) * 0.1, torch.zeros(Wd, device=dev),
                                                            torch.randn(C, Wd, 1, 1, 1, device=dev) * 0.1, torch.zeros(C, device=dev)))
        def once():
            A, B = ops.bn_fold(s, q, gamma, beta, bufs, True, N, C, S, 1e6, 1e-5, 0.1, sew, 1e6)
            (A.sum() + B.sum()).backward()
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        import time
        t0 = time.time()
        for _ in range(200):
            once()
        torch.cuda.synchronize()
        print('C=%3d se=%d: %.1f us per forward + backward (host-inclusive)' % (C, se, (time.time() - t0) / 200 * 1e6))
