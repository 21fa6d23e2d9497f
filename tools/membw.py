#!/usr/bin/env python3
"""Practical HBM ceilings on this box: device copy, fill and read-only reduction (GPU box)."""
import torch
n = 512 << 20   # floats -> 2 GiB
x = torch.empty(n, device='cuda', dtype=torch.float32).normal_()
y = torch.empty_like(x)
def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
ms = t(lambda: y.copy_(x)); print('copy  : %.1f GB/s (read+write)' % (2 * n * 4 / 1e6 / ms))
ms = t(lambda: y.fill_(1.0)); print('fill  : %.1f GB/s (write)' % (n * 4 / 1e6 / ms))
ms = t(lambda: x.sum()); print('sum   : %.1f GB/s (read)' % (n * 4 / 1e6 / ms))
ms = t(lambda: torch.add(x, 1.0, out=y)); print('add   : %.1f GB/s (read+write)' % (2 * n * 4 / 1e6 / ms))
