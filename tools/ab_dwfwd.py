#!/usr/bin/env python3
"""Same-box A/B of the depthwise forward family between two builds of the library (box-to-box spread is 5 %): sustained timing per shape
(warm clocks: 60 launches per measurement, alternating the two libraries, best of 3).

    python tools/ab_dwfwd.py libcfn_hip.so libcfn_hip_old.so"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402

LAYERS = [(54, 112, 2, 1), (54, 56, 1, 2), (108, 56, 2, 1), (108, 28, 1, 4), (216, 28, 2, 1), (216, 14, 1, 10), (432, 14, 2, 1), (432, 7, 1, 6)]


_ORIG_PROTOS = cfn_hip.header_prototypes


def use(lib):
    import ctypes
    cfn_hip._lib = None
    cfn_hip.LIB_PATH = os.path.join(ROOT, 'coarse-fine-networks_amd', 'cfn_hip', lib)
    so = ctypes.CDLL(cfn_hip.LIB_PATH)          # an older build lacks the newest entry points: bind what it has
    cfn_hip.header_prototypes = lambda path=cfn_hip.HEADER: {k: v for k, v in _ORIG_PROTOS(path).items() if hasattr(so, k)}
    cfn_hip.load()


def main():
    libs = sys.argv[1:3] if len(sys.argv) >= 3 else ['libcfn_hip.so', 'libcfn_hip_old.so']
    tot = {l: 0.0 for l in libs}
    gbt = 0.0
    for c, H, s, occ in LAYERS:
        x = torch.randn(8, c, 256, H, H, device='cuda')
        w = torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2
        A = torch.rand(8, c, device='cuda') + 0.5
        B = torch.randn(8, c, device='cuda') * 0.1
        best = {l: 1e9 for l in libs}
        for rep in range(3):
            for l in libs:
                use(l)
                for _ in range(10):
                    ops.dwconv3d(x, w, A, B, 1, s, True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(60):
                    ops.dwconv3d(x, w, A, B, 1, s, True)
                e1.record()
                torch.cuda.synchronize()
                best[l] = min(best[l], e0.elapsed_time(e1) / 60)
        Ho = H // s
        gb = 4.0 * 8 * c * 256 * (H * H + Ho * Ho) / 1e9
        gbt += gb * occ
        print('C=%3d %3d->%3d  ' % (c, H, Ho) + '   '.join('%s %7.1f us %.2f TB/s' % (l[:-3], best[l] * 1e3, gb / best[l]) for l in libs) + '  (x%d)' % occ)
        for l in libs:
            tot[l] += best[l] * occ
        del x
    print('stack: ' + '   '.join('%s %.3f ms = %.3f of 8 TB/s' % (l[:-3], tot[l], gbt / tot[l] / 8.0) for l in libs) + '  (without conv1_t)')


if __name__ == '__main__':
    main()
