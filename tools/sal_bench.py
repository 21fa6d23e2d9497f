#!/usr/bin/env python3
"""Grid Pool saliency convs (24 -> 24, 3x3x3, stride 2) at the metric's shapes: the LDS-tiled kernels of csrc/salconv.hip against
the im2col route they replace (CFN_SAL_OFF=1), forward / data gradient / weight gradient, with a result comparison.

    python tools/sal_bench.py [--batch 8] [--frames 256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=256)
    args = ap.parse_args()
    cfn_hip.load()
    g = torch.Generator(device='cpu').manual_seed(0)
    for name, T, H, pro in (('conv1 56->28', args.frames, 56, False), ('conv2 28->14', args.frames // 2, 28, True)):
        N = args.batch
        x = torch.randn(N, 24, T, H, H, generator=g).to(DEV)
        w = (torch.randn(24, 24, 3, 3, 3, generator=g) * (2.0 / 648) ** 0.5).to(DEV)
        A = (1 + 0.2 * torch.randn(N, 24, generator=g)).to(DEV) if pro else None
        B = (0.3 * torch.randn(N, 24, generator=g)).to(DEV) if pro else None
        act = 1 if pro else 0
        res = {}
        r = None
        for off in ('1', '0'):
            os.environ['CFN_SAL_OFF'] = off
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            fwd = lambda: ops.conv3d_dense(xg, wg, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)
            with torch.no_grad():
                tf = timeit(fwd)
            y, s, q = fwd()
            if r is None:
                r = torch.randn(y.shape, generator=g).to(DEV)
            loss = lambda: (y * r).sum() + (s * 0.01).sum() + (q * 0.001).sum()

            def bwd():
                xg.grad = wg.grad = None
                loss().backward(retain_graph=True)
            tb = timeit(bwd, iters=3, warm=1)
            res[off] = (tf, tb, y.detach(), s.detach(), q.detach(), xg.grad.clone(), wg.grad.clone())
        a, b = res['1'], res['0']
        gb = 4.0 * N * (24 * T * H * H + 24 * ((T - 1) // 2 + 1) * (H // 2) ** 2) / 1e9
        fl = 2.0 * N * 24 * ((T - 1) // 2 + 1) * (H // 2) ** 2 * 648 / 1e12
        rel = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
        print('%s  N=%d T=%d: forward im2col %.3f ms -> tiled %.3f ms (%.2f TB/s, %.1f TFLOP/s); fwd+bwd chain %.3f -> %.3f ms; '
              'rel diff y %.1e sum %.1e sumsq %.1e gx %.1e gw %.1e'
              % (name, N, T, a[0], b[0], gb / b[0], fl / b[0] * 1e3, a[1], b[1], rel(b[2], a[2]), rel(b[3], a[3]), rel(b[4], a[4]),
                 rel(b[5], a[5]), rel(b[6], a[6])))
        y1 = ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)[0]
        same = all(torch.equal(y1, ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)[0]) for _ in range(5))
        print('   forward bit-repeatable over 5 runs:', same)


if __name__ == '__main__':
    main()
