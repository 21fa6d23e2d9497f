#!/usr/bin/env python3
"""Device time of pws_kernel (resident weights, a wave owns all rows of its positions) against the contraction depth at fixed rows:
T(K) = a + b K estimates what a weight-STREAMING version of the same structure would cost at K = 432.  CFN_PWK=0 python tools/pws_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402


def timeit(fn, fam, reps=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    cfn_hip.prof_enable(fam, True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    cfn_hip.prof_enable(fam, False)
    ms, n, _ = cfn_hip.prof_collect(fam)
    return ms / max(n, 1) * 1e3


N, T, H = 8, 256, int(os.environ.get('H', 7))
g = torch.Generator().manual_seed(0)
for M in (192, 96):
    for K in (48, 96):
        x = torch.randn(N, K, T, H, H, generator=g).cuda()
        w = (torch.randn(M, K, 1, 1, 1, generator=g) * 0.05).cuda()
        A, B = (torch.rand(N, K, generator=g) + 0.5).cuda(), (torch.randn(N, K, generator=g) * 0.1).cuda()
        tf = timeit(lambda: ops.pwconv(x, w, A, B, 2, 1, True), 'pwconv_fwd')
        xi = torch.randn(N, M, T, H, H, generator=g).cuda().requires_grad_(True)
        wi = (torch.randn(K, M, 1, 1, 1, generator=g) * 0.05).cuda()
        y, s, q = ops.pwconv(xi, wi, None, None, 0, 1, True)
        gy = torch.randn(y.shape, generator=g).cuda()
        gs, gq = (torch.randn(s.shape, generator=g) * 0.01).cuda().to(s.dtype), (torch.randn(q.shape, generator=g) * 0.001).cuda().to(q.dtype)
        tb = timeit(lambda: torch.autograd.grad((y, s, q), (xi,), (gy, gs, gq), retain_graph=True), 'pwconv_bwd')
        print('K=%d M=%d @%dx%d: forward (swish, stats) %.1f us, data gradient (two operands) %.1f us' % (K, M, H, H, tf, tb), flush=True)
