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


def run_child(out, batch, frames):
    """one setting of CFN_SAL_OFF per PROCESS (the library reads its switches once): timings and results go to `out`"""
    cfn_hip.load()
    g = torch.Generator(device='cpu').manual_seed(0)
    res = {}
    for name, T, H, pro in (('conv1 56->28', frames, 56, False), ('conv2 28->14', frames // 2, 28, True)):
        N = batch
        x = torch.randn(N, 24, T, H, H, generator=g).to(DEV)
        w = (torch.randn(24, 24, 3, 3, 3, generator=g) * (2.0 / 648) ** 0.5).to(DEV)
        A = (1 + 0.2 * torch.randn(N, 24, generator=g)).to(DEV) if pro else None
        B = (0.3 * torch.randn(N, 24, generator=g)).to(DEV) if pro else None
        act = 1 if pro else 0
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        fwd = lambda: ops.conv3d_dense(xg, wg, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)
        with torch.no_grad():
            tf = timeit(fwd)
        y, s, q = fwd()
        r = torch.randn(y.shape, generator=g).to(DEV)
        loss = lambda: (y * r).sum() + (s * 0.01).sum() + (q * 0.001).sum()

        def bwd():
            xg.grad = wg.grad = None
            loss().backward(retain_graph=True)
        tb = timeit(bwd, iters=3, warm=1)
        y1 = ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)[0]
        same = all(torch.equal(y1, ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)[0]) for _ in range(5))
        res[name] = dict(tf=tf, tb=tb, same=same, N=N, T=T, H=H, y=y.detach().cpu(), s=s.detach().cpu(), q=q.detach().cpu(),
                         gx=xg.grad.cpu(), gw=wg.grad.cpu())
    torch.save(res, out)


def main():
    import subprocess
    import tempfile
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--child', default=None)
    args = ap.parse_args()
    if args.child:
        return run_child(args.child, args.batch, args.frames)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for off in ('1', '0'):
            out = os.path.join(td, 'sal_%s.pt' % off)
            env = dict(os.environ, CFN_SAL_OFF=off)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--batch', str(args.batch), '--frames', str(args.frames), '--child', out],
                           env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            res[off] = torch.load(out)
    rel = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
    for name in res['0']:
        a, b = res['1'][name], res['0'][name]
        N, T, H = b['N'], b['T'], b['H']
        gb = 4.0 * N * (24 * T * H * H + 24 * ((T - 1) // 2 + 1) * (H // 2) ** 2) / 1e9
        fl = 2.0 * N * 24 * ((T - 1) // 2 + 1) * (H // 2) ** 2 * 648 / 1e12
        print('%s  N=%d T=%d: forward im2col %.3f ms -> tiled %.3f ms (%.2f TB/s, %.1f TFLOP/s); fwd+bwd chain %.3f -> %.3f ms; '
              'rel diff y %.1e sum %.1e sumsq %.1e gx %.1e gw %.1e'
              % (name, N, T, a['tf'], b['tf'], gb / b['tf'], fl / b['tf'] * 1e3, a['tb'], b['tb'], rel(b['y'], a['y']), rel(b['s'], a['s']),
                 rel(b['q'], a['q']), rel(b['gx'], a['gx']), rel(b['gw'], a['gw'])))
        print('   forward bit-repeatable over 5 runs:', b['same'])


if __name__ == '__main__':
    main()
