#!/usr/bin/env python3
"""Run-to-run stress of the pointwise weight-gradient entry point on the X3D-M layer shapes: `--runs` launches per shape on identical
inputs, every result compared with the first.  fp64 atomics make differences of ~1e-16 legitimate; anything larger is a race.

    python tools/wgrad_stress.py [--runs 300] [--frames 16] [--batch 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402

DEV = 'cuda'
# name, Cin (K), Cout (M), plane, act of the prologue
SHAPES = [('L2 conv3 108->48 @28', 108, 48, 28, 2), ('L2 conv1 48->108 @28', 48, 108, 28, 0), ('L3 conv1 96->216 @14', 96, 216, 14, 0),
          ('L3 conv3 216->96 @14', 216, 96, 14, 2), ('L4 conv1 192->432 @7', 192, 432, 7, 0), ('L4 conv3 432->192 @7', 432, 192, 7, 2),
          ('L1 conv3 54->24 @56', 54, 24, 56, 2), ('L1 conv1 24->54 @56', 24, 54, 56, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--runs', type=int, default=300)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--only', default=None)
    ap.add_argument('--pollute', type=int, default=0)
    a = ap.parse_args()
    cfn_hip.load()
    g = torch.Generator().manual_seed(0)
    bad_total = 0
    for name, K, M, H, act in SHAPES:
        if a.only and a.only not in name:
            continue
        N, T = a.batch, a.frames
        gy = torch.randn(N, M, T, H, H, generator=g).to(DEV)
        y = torch.randn(N, M, T, H, H, generator=g).to(DEV)
        x = torch.randn(N, K, T, H, H, generator=g).to(DEV)
        gs = (0.05 * torch.randn(N, M, generator=g)).double().to(DEV)
        gq = (0.01 * torch.randn(N, M, generator=g)).double().to(DEV)
        A = (1 + 0.2 * torch.randn(N, K, generator=g)).double().to(DEV)
        B = (0.2 * torch.randn(N, K, generator=g)).double().to(DEV)
        ref = None
        bad, worst = 0, 0.0
        # --pollute: a different kernel with different data runs in front of every launch (what LDS / registers / caches hold when the
        # weight-gradient kernel starts then differs from launch to launch, as it does inside a training step)
        junk_x = torch.randn(N, K, T, H, H, generator=g).to(DEV)
        junk_w = (0.1 * torch.randn(M, K, generator=g)).to(DEV)
        for r in range(a.runs):
            if a.pollute:
                junk_x.mul_(1.0001).add_(0.001 * r)
                jy = torch.empty(N, M, T, H, H, device=DEV)
                js, jq = (torch.zeros(N, M, dtype=torch.float64, device=DEV) for _ in range(2))
                cfn_hip.call('cfn_pwconv_fwd', junk_x, A, B, act, junk_w, jy, js, jq, N, K, M, T, H, H, 1)
                jgx = torch.empty_like(junk_x)
                ja, jb = (torch.zeros(N, K, dtype=torch.float64, device=DEV) for _ in range(2))
                cfn_hip.call('cfn_pwconv_bwd_data', jy, jy, gs, gq, junk_w, junk_x, A, B, act, jgx, ja, jb, N, K, M, T, H, H, 1)
            if a.pollute >= 2:      # a weight gradient of ANOTHER layer shape (another template variant, other LDS layout) on changing data
                if r == 0:
                    j2 = [torch.randn(N, 216, T, 14, 14, generator=g).to(DEV), torch.randn(N, 216, T, 14, 14, generator=g).to(DEV),
                          torch.randn(N, 96, T, 14, 14, generator=g).to(DEV), torch.randn(N, 216, generator=g).double().to(DEV),
                          torch.randn(N, 96, generator=g).double().to(DEV)]
                j2[0].mul_(1.0001).add_(0.001 * r)
                j2[2].mul_(0.9999).add_(0.002 * r)
                jw = torch.zeros(216, 96, dtype=torch.float64, device=DEV)
                cfn_hip.call('cfn_pwconv_bwd_weight', j2[0], j2[1], j2[3], j2[3], j2[2], j2[4], j2[4], 0, jw, N, 96, 216, T, 14, 14, 1, None)
            gw = torch.zeros(M, K, dtype=torch.float64, device=DEV)
            cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, act, gw, N, K, M, T, H, H, 1, None)
            if ref is None:
                ref = gw
                continue
            d = float((gw - ref).abs().max() / ref.abs().max())
            if d > 1e-12:
                bad += 1
                worst = max(worst, d)
        print('%-26s %d of %d launches differ by more than 1e-12 (worst %.2e of max |gw|)' % (name, bad, a.runs - 1, worst))
        bad_total += bad
    print('TOTAL', bad_total)


if __name__ == '__main__':
    main()
