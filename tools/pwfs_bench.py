#!/usr/bin/env python3
"""One-pass split-bf16 pointwise backward (csrc/pwfuseds.hip, CFN_PWF_SPLIT=1) against the separate data / weight gradient kernels at the
layer-2 shapes of the benchmark (8 clips x 256 frames x 28 x 28), through the C ABI.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402

DEV = 'cuda'
N, T, H0 = int(os.environ.get('NB', '8')), int(os.environ.get('FRAMES', '256')), int(os.environ.get('HW', '28'))
CASES = [('L1 conv1 24->54 @2H (no prologue)', 24, 54, None), ('L1 conv3 54->24 @2H (swish prologue)', 54, 24, 2), ('L1 conv1 24->54 @4H (no prologue)', 24, 54, None),
         ] if os.environ.get('L1') else [('L3 conv1 96->216 @H/2 (no prologue)', 96, 216, None), ('L3.0 conv1 48->216 @H + compact shortcut gradient', 48, 216, None), ('L3 conv3 216->96 @H/2 (swish prologue)', 216, 96, 2)] if os.environ.get('L3') else [('conv1 48->108 (no prologue)', 48, 108, None), ('conv1 24->108 @2H (no prologue)', 24, 108, None), ('conv1 24->108 @2H + compact shortcut gradient', 24, 108, None), ('conv3 108->48 (swish prologue)', 108, 48, 2), ('conv1 48->108 (relu prologue)', 48, 108, 1)]


def timeit(fn, iters=10, warm=3):
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


for name, Cin, Cout, act in CASES:
    H = (4 * H0) if '@4H' in name else (2 * H0) if '@2H' in name else (H0 // 2) if '@H/2' in name else H0
    g = torch.Generator().manual_seed(1)
    gy, y = torch.randn(N, Cout, T, H, H, generator=g).to(DEV), torch.randn(N, Cout, T, H, H, generator=g).to(DEV)
    x = torch.randn(N, Cin, T, H, H, generator=g).to(DEV)
    w = (0.3 * torch.randn(Cout, Cin, generator=g)).to(DEV)
    f64 = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).double().to(DEV)
    gs, gq, gsc = f64(N, Cout, scale=0.05), f64(N, Cout, scale=0.01), 1.0 + f64(N, Cout, scale=0.3)
    A = B = gA = gB = None
    if act is not None:
        A, B = 1.0 + f64(N, Cin, scale=0.2), f64(N, Cin, scale=0.2)
        gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
    gx, gw = torch.empty_like(x), torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
    a_ = 0 if act is None else act
    accg = torch.randn(N, Cin, T, H // 2, H // 2, generator=g).to(DEV) if 'shortcut' in name else None
    accs = 2 if accg is not None else 1

    def separate():
        cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, N, Cin, Cout, T, H, H, 1, accg, accs, gsc)
        cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, a_, gw, N, Cin, Cout, T, H, H, 1, gsc)

    def fused():
        assert cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, gw, N, Cin, Cout, T, H, H, accg, accs, gsc)

    os.environ['CFN_PWF_SPLIT'] = '0'
    os.environ['CFN_PWF_L3E'] = '1' if os.environ.get('L3') else '0'
    ts = timeit(separate)
    tl1 = timeit(fused) if os.environ.get('L1') else None      # the fp32 fused kernel of pwfused.hip
    os.environ['CFN_PWF_SPLIT'] = '1' if os.environ.get('L3') else '3' if os.environ.get('L1') else '2'
    tf = timeit(fused)
    Q = N * T * H * H
    floor = 4.0 * Q * (2 * Cout + 2 * Cin) / 1e9          # gy, y, x read once, gx written once
    if tl1 is not None:
        print('%-40s fp32 fused kernel (pwfused.hip) %.3f ms' % (name, tl1))
    print('%-32s separate %.3f ms   fused (split bf16) %.3f ms   4-pass traffic %.2f GB = %.2f TB/s fused' % (name, ts, tf, floor, floor / tf))
