#!/usr/bin/env python3
"""Step time of the dispatcher route (CFN_USE_TORCH_OPS=1: every block through torch.ops.cfn.*, native TORCH_LIBRARY operators, plain semantics) against
the default route (cfn_hip.ops autograd Functions over ctypes, with the cross-operator fusions: deferred prologues between blocks, shortcut tokens,
tail links, one gradient cast per pass) -- VERDICT r5 next-step 8: "so the claim why the default bypasses the dispatcher has a number".

    python tools/torchops_route.py [--batch 8] [--frames 256] [--steps 5] > profiles/r06_torchops_route.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import torch.optim as optim       # noqa: E402
import x3d_fine                   # noqa: E402
import train_fine                 # noqa: E402
from cfn_hip import dist as cdist  # noqa: E402


def run(route, B, T, steps):
    x3d_fine.USE_TORCH_OPS = route == 'dispatcher'
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = train_fine.build_model(dev, pretrained=None).train(True)
    opt = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
    red = cdist.GradReducer(net.parameters())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, T, 224, 224, generator=g).to(dev)
    labels = (torch.rand(B, 157, T * 10, generator=g) < 0.05).float().to(dev)
    masks = torch.ones(B, T * 10, device=dev)
    losses = []
    for _ in range(2):
        losses.append(train_fine.train_step(net, red, opt, x, labels, masks)[:2])
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(train_fine.train_step(net, red, opt, x, labels, masks)[:2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    red.close()
    return dt * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30, [float(v) for v in losses[0]], [float(v) for v in losses[-1]]


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--steps', type=int, default=5)
    a = ap.parse_args()
    import cfn_hip.torchlib as tl
    print('# x3d_fine X3D-M train step, %d x 3 x %d x 224 x 224, fp32; native operator library: %s' % (a.batch, a.frames, tl.NATIVE))
    for route in ('ctypes', 'dispatcher', 'ctypes', 'dispatcher'):
        ms, gb, l0, l1 = run(route, a.batch, a.frames, a.steps)
        print('%-10s %8.2f ms/step   peak %6.1f GiB   first loss (cls, loc) %s   last %s' % (route, ms, gb, ['%.5f' % v for v in l0], ['%.5f' % v for v in l1]), flush=True)
        torch.cuda.empty_cache()
