#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache pay for the second read of a pointwise conv's backward?  (GPU box only)

The data gradient and the weight gradient of a 1x1x1 conv read the same three tensors (gy, y, x).  Run over the whole 8-clip batch (2M + K rows x
401k positions x 4 B = 0.85 GB at layer 3) the second kernel finds nothing of the first one's reads on the die; run clip by clip (106 MB per clip)
it could.  This probe times both orders with the product's own kernels through autograd: (a) one backward over 8 clips, (b) eight backwards over
one clip each (data gradient of clip i, weight gradient of clip i, ...), (c) 4 x 2 clips."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402
from microbench import devtime, timeit, DEV      # noqa: E402

T, NB = 256, 8
SHAPES = [('L2 conv1 48->108 @28', 48, 108, 28), ('L2 conv3 108->48 @28', 108, 48, 28), ('L3 conv1 96->216 @14', 96, 216, 14),
          ('L3 conv3 216->96 @14', 216, 96, 14), ('L4 conv1 192->432 @7', 192, 432, 7), ('L4 conv3 432->192 @7', 432, 192, 7)]

for name, ci, co, H in SHAPES:
    x = torch.randn(NB, ci, T, H, H, device=DEV)
    w = (torch.randn(co, ci, 1, 1, 1, device=DEV) * 0.1).requires_grad_(True)
    A = torch.rand(NB, ci, device=DEV) + 0.5
    B = torch.randn(NB, ci, device=DEV) * 0.1
    gy_all = torch.randn(NB, co, T, H, H, device=DEV)
    row = '%-24s' % name
    for k in (8, 2, 1):
        parts = []
        for i in range(0, NB, k):
            xr = x[i:i + k].clone().requires_grad_(True)
            Ar, Br = A[i:i + k].clone().requires_grad_(True), B[i:i + k].clone().requires_grad_(True)
            y, sm, sq = ops.pwconv(xr, w, Ar, Br, 2, 1, True)
            gs, gq = torch.randn_like(sm) * 0.01, torch.randn_like(sq) * 0.001
            parts.append(((y, sm, sq), (xr, w, Ar, Br), (gy_all[i:i + k].contiguous(), gs, gq)))

        def f():
            for outs, ins, gos in parts:
                torch.autograd.grad(outs, ins, gos, retain_graph=True)
        wall = timeit(f, iters=5, warm=2)
        d = devtime(f)
        row += ' | %d x %d clips: wall %.3f ms, dgrad %.3f + wgrad %.3f' % (NB // k, k, wall, d.get('pwconv_bwd', 0.0), d.get('pwconv_wgrad', 0.0))
        del parts
    print(row, flush=True)
    del x, gy_all
