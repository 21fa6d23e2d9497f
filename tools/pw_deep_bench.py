#!/usr/bin/env python3
"""Device time of the layer-4 deep contractions (K = 432) at the benchmark's size: forward 432 -> 192 (Swish prologue, statistics) and the
data gradient of a 192 -> 432 conv (contraction over its 432 outputs, g' = gy + gs + 2 gq y).  Run once per setting of CFN_PWT
(0 = pw_deep_kernel, 3 = pws_kernel with streamed weights) on the same box:  for v in 0 3; do CFN_PWT=$v python tools/pw_deep_bench.py; done"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
if os.environ.get('CFN_LIB'):
    cfn_hip.LIB_PATH = os.path.join(ROOT, os.environ['CFN_LIB'])
from cfn_hip import ops           # noqa: E402

DEV = 'cuda'


def timeit(fn, fam, reps=40):
    """device time per launch from the HIP events the C ABI records around its launches (the Python call costs as much as the kernel)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    cfn_hip.prof_enable(fam, True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    cfn_hip.prof_enable(fam, False)
    ms, n, _ = cfn_hip.prof_collect(fam)
    return ms / max(n, 1) * 1e3


def main():
    N, T = int(os.environ.get('N', 8)), int(os.environ.get('T', 256))
    g = torch.Generator().manual_seed(0)
    for (K, M, H) in [(432, 192, 7), (432, 432, 7), (432, 96, 14)][:int(os.environ.get('SHAPES', 3))]:
        x = torch.randn(N, K, T, H, H, generator=g).to(DEV)
        w = (torch.randn(M, K, 1, 1, 1, generator=g) * 0.05).to(DEV)
        A, B = (torch.rand(N, K, generator=g) + 0.5).to(DEV), (torch.randn(N, K, generator=g) * 0.1).to(DEV)
        t_f = timeit(lambda: ops.pwconv(x, w, A, B, 2, 1, True), 'pwconv_fwd')
        xr, Ar, Br = x.clone().requires_grad_(True), A.clone().requires_grad_(True), B.clone().requires_grad_(True)
        yf, sf, qf = ops.pwconv(xr, w, Ar, Br, 2, 1, True)
        gyf = torch.randn(yf.shape, generator=g).to(DEV)
        gsf, gqf = (torch.randn(sf.shape, generator=g) * 0.01).to(DEV).to(sf.dtype), (torch.randn(qf.shape, generator=g) * 0.001).to(DEV).to(qf.dtype)
        t_e = timeit(lambda: torch.autograd.grad((yf, sf, qf), (xr, Ar, Br), (gyf, gsf, gqf), retain_graph=True), 'pwconv_bwd')
        del xr, yf, gyf
        xi = torch.randn(N, M, T, H, H, generator=g).to(DEV).requires_grad_(True)
        wi = (torch.randn(K, M, 1, 1, 1, generator=g) * 0.05).to(DEV)
        y, s, q = ops.pwconv(xi, wi, None, None, 0, 1, True)
        gy = torch.randn(y.shape, generator=g).to(DEV)
        gs, gq = (torch.randn(s.shape, generator=g) * 0.01).to(DEV).to(s.dtype), (torch.randn(q.shape, generator=g) * 0.001).to(DEV).to(q.dtype)
        t_b2 = timeit(lambda: torch.autograd.grad((y, s, q), (xi,), (gy, gs, gq), retain_graph=True), 'pwconv_bwd')
        t_b1 = timeit(lambda: torch.autograd.grad((y, s), (xi,), (gy, gs), retain_graph=True), 'pwconv_bwd')
        print('CFN_PWT=%s  K=%d M=%d @%dx%d N=%d T=%d: forward %.1f us [its data gradient with the act-prime epilogue, %d -> %d rows: %.1f us], data gradient (two operands) %.1f us, (one operand) %.1f us'
              % (os.environ.get('CFN_PWT', 'default') + ' ' + os.path.basename(os.environ.get('CFN_LIB', '')), K, M, H, H, N, T, t_f, M, K, t_e, t_b2, t_b1), flush=True)
        del x, w, xi, wi, y, gy


if __name__ == '__main__':
    main()
