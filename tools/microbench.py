#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the X3D-M layer shapes (GPU box only).

    python tools/microbench.py [pw|dw|all] [--frames 256] [--bwd]

Times each op with HIP events on torch's current stream (the stream the C ABI launches on) and prints
algorithmic GB/s (N_in + N_out elements x 4 B per pass, as SURVEY 8d counts them) and TFLOP/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402

DEV = 'cuda'


FAMS = ('dwconv_fwd', 'dwconv_bwd', 'dwconv_wgrad', 'pwconv_fwd', 'pwconv_bwd', 'pwconv_wgrad')


def devtime(fn, iters=3):
    """device ms per call of each kernel family (HIP events around every launch inside the C ABI)"""
    fn()
    torch.cuda.synchronize()
    for f in FAMS:
        cfn_hip.prof_enable(f, True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    out = {}
    for f in FAMS:
        cfn_hip.prof_enable(f, False)
        ms, n, _ = cfn_hip.prof_collect(f)
        if n:
            out[f] = ms / iters
    return out


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


# (name, Cin, Cout, H(in), stride)
PW_LAYERS = [('L1.0 conv1 24->54 @112', 24, 54, 112, 1), ('L1.0 ds 24->24 s2', 24, 24, 112, 2),
             ('L1.x conv3 54->24 @56', 54, 24, 56, 1), ('L1.x conv1 24->54 @56', 24, 54, 56, 1),
             ('L2.0 conv1 24->108 @56', 24, 108, 56, 1), ('L2.x conv3 108->48 @28', 108, 48, 28, 1),
             ('L2.x conv1 48->108 @28', 48, 108, 28, 1), ('L3.x conv1 96->216 @14', 96, 216, 14, 1),
             ('L3.x conv3 216->96 @14', 216, 96, 14, 1), ('L4.x conv1 192->432 @7', 192, 432, 7, 1),
             ('L4.x conv3 432->192 @7', 432, 192, 7, 1)]
DW_LAYERS = [('L1.0 dw 54 112->56 s2', 54, 112, 2), ('L1.x dw 54 @56', 54, 56, 1), ('L2.0 dw 108 56->28 s2', 108, 56, 2),
             ('L2.x dw 108 @28', 108, 28, 1), ('L3.0 dw 216 28->14 s2', 216, 28, 2), ('L3.x dw 216 @14', 216, 14, 1),
             ('L4.0 dw 432 14->7 s2', 432, 14, 2), ('L4.x dw 432 @7', 432, 7, 1)]


def bench_pw(T, bwd, NB=1, only=None):
    print('%-28s %9s %9s %9s %s' % ('pointwise layer', 'ms', 'GB/s', 'TFLOP/s', '(fwd)'))
    for name, ci, co, H, s in PW_LAYERS:
        if only and only not in name:
            continue
        x = torch.randn(NB, ci, T, H, H, device=DEV)
        w = torch.randn(co, ci, 1, 1, 1, device=DEV) * 0.1
        A = torch.rand(NB, ci, device=DEV) + 0.5
        B = torch.randn(NB, ci, device=DEV) * 0.1
        Ho = (H - 1) // s + 1
        Q = NB * T * Ho * Ho
        ms = devtime(lambda: ops.pwconv(x, w, A, B, 2, s, True))['pwconv_fwd']
        gb = 4.0 * (ci * Q + co * Q) / 1e9
        print('%-28s %9.3f %9.1f %9.2f' % (name, ms, gb / ms * 1e3, 2.0 * ci * co * Q / ms / 1e9))
        if bwd:
            xr = x.clone().requires_grad_(True)
            wr = w.clone().requires_grad_(True)
            Ar, Br = A.clone().requires_grad_(True), B.clone().requires_grad_(True)
            y, sm, sq = ops.pwconv(xr, wr, Ar, Br, 2, s, True)
            gy, gs, gq = torch.randn_like(y), torch.randn_like(sm) * 0.01, torch.randn_like(sq) * 0.001

            def f():
                torch.autograd.grad((y, sm, sq), (xr, wr, Ar, Br), (gy, gs, gq), retain_graph=True)
            d = devtime(f)
            md, mw = d.get('pwconv_bwd', 0.0), d.get('pwconv_wgrad', 0.0)
            if mw == 0.0:
                print('%-28s dgrad+wgrad fused %7.3f ms %7.1f GB/s (4 passes) %6.2f TF' %
                      ('', md, 4.0 * (2 * ci + 2 * co) * Q / 1e6 / md, 4.0 * ci * co * Q / md / 1e9))
                continue
            print('%-28s dgrad %7.3f ms %7.1f GB/s %6.2f TF | wgrad %7.3f ms %7.1f GB/s %6.2f TF' %
                  ('', md, 4.0 * (2 * ci + 2 * co) * Q / 1e6 / md, 2.0 * ci * co * Q / md / 1e9,
                   mw, 4.0 * (ci + 2 * co) * Q / 1e6 / mw, 2.0 * ci * co * Q / mw / 1e9))


def bench_dw(T, bwd, NB=1):
    print('%-28s %9s %9s %9s %s' % ('depthwise layer', 'ms', 'GB/s', 'TFLOP/s', '(fwd)'))
    x = torch.randn(NB, 24, T, 112, 112, device=DEV)
    w = torch.randn(24, 1, 5, 1, 1, device=DEV) * 0.3
    ms = devtime(lambda: ops.dwconv_t5(x, w, True))['dwconv_fwd']
    print('%-28s %9.3f %9.1f' % ('stem conv1_t 24 @112 (5x1x1)', ms, 8.0 * NB * 24 * T * 112 * 112 / 1e6 / ms))
    if bwd:
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y, sm, sq = ops.dwconv_t5(xr, wr, True)
        gy, gs, gq = torch.randn_like(y), torch.randn_like(sm) * 0.01, torch.randn_like(sq) * 0.001
        md = devtime(lambda: torch.autograd.grad((y, sm, sq), (xr, wr), (gy, gs, gq), retain_graph=True))['dwconv_bwd']
        print('%-28s dgrad+wgrad %7.3f ms %7.1f GB/s' % ('', md, 28.0 * NB * 24 * T * 112 * 112 / 1e6 / md))
        del xr, y, gy
    del x
    for name, c, H, s in DW_LAYERS:
        x = torch.randn(NB, c, T, H, H, device=DEV)
        w = torch.randn(c, 1, 3, 3, 3, device=DEV) * 0.2
        A = torch.rand(NB, c, device=DEV) + 0.5
        B = torch.randn(NB, c, device=DEV) * 0.1
        Ho = (H + 2 - 3) // s + 1
        ms = devtime(lambda: ops.dwconv3d(x, w, A, B, 1, s, True))['dwconv_fwd']
        gb = 4.0 * NB * c * T * (H * H + Ho * Ho) / 1e9
        print('%-28s %9.3f %9.1f %9.2f' % (name, ms, gb / ms * 1e3, 54.0 * NB * c * T * Ho * Ho / ms / 1e9))
        if bwd:
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            Ar, Br = A.clone().requires_grad_(True), B.clone().requires_grad_(True)
            y, sm, sq = ops.dwconv3d(xr, wr, Ar, Br, 1, s, True)
            gy, gs, gq = torch.randn_like(y), torch.randn_like(sm) * 0.01, torch.randn_like(sq) * 0.001

            def f():
                torch.autograd.grad((y, sm, sq), (xr, wr, Ar, Br), (gy, gs, gq), retain_graph=True)
            d = devtime(f)
            md, mw = d.get('dwconv_bwd', 0.0), d.get('dwconv_wgrad', 0.0)
            if mw == 0.0:      # fused data + weight gradient: one kernel, 4 tensor passes
                # bytes: x read + gx written at input resolution, gy + y read at output resolution
                print('%-28s dgrad+wgrad fused %7.3f ms %7.1f GB/s (x, gx, gy, y once each)' %
                      ('', md, 4.0 * NB * c * T * (2 * H * H + 2 * Ho * Ho) / 1e6 / md))
            else:
                print('%-28s dgrad %7.3f ms %7.1f GB/s | wgrad %7.3f ms %7.1f GB/s' %
                      ('', md, 4.0 * NB * c * T * (2 * H * H + 2 * Ho * Ho) / 1e6 / md, mw, 4.0 * NB * c * T * (H * H + 2 * Ho * Ho) / 1e6 / mw))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('what', nargs='?', default='all')
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--bwd', action='store_true')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    if a.what in ('pw', 'all'):
        bench_pw(a.frames, a.bwd, a.batch, a.only)
    if a.what in ('dw', 'all'):
        bench_dw(a.frames, a.bwd, a.batch)
