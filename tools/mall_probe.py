#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache serve a consumer kernel that runs right behind its producer?  Depthwise forward shapes of X3D-M (8 clips x 256
frames), timed one launch at a time (HIP events) (a) cold: a 2 GB fill in front, (b) hot: `x.copy_(src)` (the producer: writes x front to
back) directly in front, with the library as built (every XCD walks its eighth of the tensor front to back) and with a build whose XCD remap
walks back to front (-DCFN_XCD_REVERSE on dwflat.hip / dwsmall.hip: what was written last is read first)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402
from cfn_hip import ops           # noqa: E402

LAYERS = [(432, 7, 1), (432, 14, 2), (216, 14, 1), (216, 28, 2), (108, 28, 1), (54, 56, 1)]


def use(lib):
    cfn_hip._lib = None
    cfn_hip.LIB_PATH = os.path.join(ROOT, 'coarse-fine-networks_amd', 'cfn_hip', lib)
    cfn_hip.load()


def main():
    junk = torch.empty(512 << 20, device='cuda')
    for c, H, s in LAYERS:
        x = torch.randn(8, c, 256, H, H, device='cuda')
        src = x.clone()
        w = torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2
        A = torch.rand(8, c, device='cuda') + 0.5
        B = torch.randn(8, c, device='cuda') * 0.1
        Ho = H // s
        gb = 4.0 * 8 * c * 256 * (H * H + Ho * Ho) / 1e9
        res = {}
        for lib in ('libcfn_hip.so', 'libcfn_hip_rev.so'):
            use(lib)
            for mode in ('cold', 'hot'):
                ts = []
                for it in range(12):
                    if mode == 'cold':
                        x.copy_(src); junk.fill_(float(it))
                    else:
                        junk.fill_(float(it)); x.copy_(src)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.dwconv3d(x, w, A, B, 1, s, True)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                ts = sorted(ts[2:])
                res[(lib, mode)] = ts[len(ts) // 2]
        print('C=%3d %3d->%3d (%4.0f MB in): forward walk cold %7.1f us  hot %7.1f us | reverse walk cold %7.1f  hot %7.1f   (%.2f / %.2f / %.2f / %.2f TB/s)'
              % (c, H, Ho, 4.0 * 8 * c * 256 * H * H / 1e6, res[('libcfn_hip.so', 'cold')], res[('libcfn_hip.so', 'hot')], res[('libcfn_hip_rev.so', 'cold')],
                 res[('libcfn_hip_rev.so', 'hot')], *[gb / (res[k] * 1e-6) / 1e3 for k in (('libcfn_hip.so', 'cold'), ('libcfn_hip.so', 'hot'), ('libcfn_hip_rev.so', 'cold'), ('libcfn_hip_rev.so', 'hot'))]))
        del x, src


if __name__ == '__main__':
    main()
