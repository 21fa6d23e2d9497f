#!/usr/bin/env python3
"""A/B of two builds of the library in ONE process on ONE box (box-to-box spread is 5-10 %): run-to-run determinism of the
pointwise forward / data gradient (no atomics on those outputs: they must repeat bit for bit) and per-layer device times.

    python tools/ab_pw.py [--libs libcfn_hip.so libcfn_hip_old.so] [--reps 200] [--batch 8] [--frames 256]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402

DEV = 'cuda'
FAMS = ('pwconv_fwd', 'pwconv_bwd', 'pwconv_wgrad')
LAYERS = [('L2 conv3 108->48 @28', 108, 48, 28), ('L2 conv1 48->108 @28', 48, 108, 28), ('L3 conv1 96->216 @14', 96, 216, 14),
          ('L3 conv3 216->96 @14', 216, 96, 14), ('L4 conv1 192->432 @7', 192, 432, 7), ('L4 conv3 432->192 @7', 432, 192, 7)]


def use(lib):
    cfn_hip._lib = None
    cfn_hip.LIB_PATH = os.path.join(ROOT, 'coarse-fine-networks_amd', 'cfn_hip', lib)
    cfn_hip.load()


def setup(N, ci, co, T, H, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(N, ci, T, H, H, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, 1, generator=g) * 0.1).to(DEV).requires_grad_(True)
    A = (torch.rand(N, ci, generator=g) + 0.5).to(DEV).requires_grad_(True)
    B = (torch.randn(N, ci, generator=g) * 0.1).to(DEV).requires_grad_(True)
    return x, w, A, B


def once(x, w, A, B, gyy=None):
    y, sm, sq = ops.pwconv(x, w, A, B, 2, 1, True)
    if gyy is None:
        g = torch.Generator(device='cpu').manual_seed(1)
        gyy = (torch.randn(y.shape, generator=g).to(DEV), (torch.randn(sm.shape, generator=g) * 0.01).to(DEV).double(),
               (torch.randn(sq.shape, generator=g) * 0.001).to(DEV).double())
    gx, gw, gA, gB = torch.autograd.grad((y, sm, sq), (x, w, A, B), gyy)
    return (y.detach(), gx, gw, sm.detach()), gyy


def determinism(reps):
    for N, ci, co, T, H in [(2, 216, 96, 2, 14), (2, 96, 216, 2, 14), (1, 108, 48, 3, 6), (2, 108, 48, 4, 28), (2, 48, 108, 5, 28)]:
        args = setup(N, ci, co, T, H)
        ref, gyy = once(*args)
        bad = [0, 0]
        worst = 0.0
        for _ in range(reps):
            out, _ = once(*args, gyy=gyy)
            for i in (0, 1):
                if not torch.equal(out[i], ref[i]):
                    bad[i] += 1
                    worst = max(worst, float((out[i] - ref[i]).abs().max() / ref[i].abs().max()))
        print('  determinism %3d->%3d N=%d T=%d H=%d: y differs in %d / %d runs, gx in %d / %d (worst rel %.2e)'
              % (ci, co, N, T, H, bad[0], reps, bad[1], reps, worst))


def timing(NB, T):
    for name, ci, co, H in LAYERS:
        args = setup(NB, ci, co, T, H)
        _, gyy = once(*args)
        torch.cuda.synchronize()
        for f in FAMS:
            cfn_hip.prof_enable(f, True)
        for _ in range(5):
            once(*args, gyy=gyy)
        torch.cuda.synchronize()
        t = {}
        for f in FAMS:
            cfn_hip.prof_enable(f, False)
            ms, n, _b = cfn_hip.prof_collect(f)
            t[f] = ms / 5 if n else 0.0
        print('  %-24s fwd %.3f  dgrad %.3f  wgrad %.3f ms' % (name, t['pwconv_fwd'], t['pwconv_bwd'], t['pwconv_wgrad']))
        del args, gyy


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--libs', nargs='+', default=['libcfn_hip.so', 'libcfn_hip_old.so'])
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=2)
    a = ap.parse_args()
    for r in range(a.rounds):
        for lib in a.libs:
            if not os.path.exists(os.path.join(ROOT, 'coarse-fine-networks_amd', 'cfn_hip', lib)):
                continue
            use(lib)
            print('%s (round %d)' % (lib, r))
            if r == 0 and a.reps:
                determinism(a.reps)
            timing(a.batch, a.frames)
