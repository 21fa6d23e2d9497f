#!/usr/bin/env python3
"""Per-shape timing of the depthwise 3x3x3 forward entry point (the conv2 shapes of X3D-M at 224x224 input): HIP events around `--reps`
launches per shape, algorithmic TB/s (input + output tensor once).  Same-box A/B of kernel generations through the environment:

    CFN_DW_FLAT=0 python tools/dwfwd_shapes.py        # column-pair wave kernels (dwcp.hip) everywhere
    python tools/dwfwd_shapes.py                      # flat kernels (dwflat.hip) where they are dispatched
    python tools/dwfwd_shapes.py --only 56 --frames 256 --batch 8"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
from cfn_hip import ops           # noqa: E402

# (channels, H_in, stride, occurrences in X3D-M)
LAYERS = [(54, 112, 2, 1), (54, 56, 1, 2), (108, 56, 2, 1), (108, 28, 1, 4), (216, 28, 2, 1), (216, 14, 1, 10), (432, 14, 2, 1), (432, 7, 1, 6)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', type=int, default=0, help='only this input plane size')
    ap.add_argument('--stride', type=int, default=0)
    a = ap.parse_args()
    tot = 0.0
    for c, H, s, occ in LAYERS:
        if (a.only and H != a.only) or (a.stride and s != a.stride):
            continue
        x = torch.randn(a.batch, c, a.frames, H, H, device='cuda')
        w = torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2
        A = torch.rand(a.batch, c, device='cuda') + 0.5
        B = torch.randn(a.batch, c, device='cuda') * 0.1
        for _ in range(2):
            ops.dwconv3d(x, w, A, B, 1, s, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(a.reps):
                ops.dwconv3d(x, w, A, B, 1, s, True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / a.reps)
        Ho = H // s
        gb = 4.0 * a.batch * c * a.frames * (H * H + Ho * Ho) / 1e9
        # (includes the allocation of the output and the two statistics tensors by the Python wrapper: a few us)
        print('C=%3d %3d->%3d  %8.1f us  %.2f TB/s  (x%d per step)' % (c, H, Ho, best * 1e3, gb / best, occ))
        tot += best * occ
        del x
    print('stack total %.3f ms' % tot)


if __name__ == '__main__':
    main()
