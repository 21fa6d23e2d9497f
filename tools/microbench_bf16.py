#!/usr/bin/env python3
"""Device times of the bf16 pointwise kernels at X3D-M layer shapes (T frames, B clips; env T, B), through the C ABI.

    python tools/microbench_bf16.py            # fwd / dgrad (with and without act' epilogue) / wgrad per layer shape"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import cfn_hip
from cfn_hip import call
T = int(os.environ.get('T', '256')); NB = int(os.environ.get('B', '8'))
dev = 'cuda'
BF = torch.bfloat16


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# (name, Cin, Cout, H)
SHAPES = [('L1 conv1 24->54 @56', 24, 54, 56), ('L1 conv3 54->24 @56', 54, 24, 56), ('L2 conv1 48->108 @28', 48, 108, 28),
          ('L2 conv3 108->48 @28', 108, 48, 28), ('L3 conv1 96->216 @14', 96, 216, 14), ('L3 conv3 216->96 @14', 216, 96, 14),
          ('L4 conv1 192->432 @7', 192, 432, 7), ('L4 conv3 432->192 @7', 432, 192, 7)]
only = sys.argv[1:] 
for name, ci, co, H in SHAPES:
    if only and not any(o in name for o in only):
        continue
    Q = T * H * H
    x = torch.randn(NB, ci, Q, device=dev).to(BF)
    y = torch.randn(NB, co, Q, device=dev).to(BF)
    gy = torch.randn(NB, co, Q, device=dev).to(BF)
    gx = torch.empty_like(x)
    w = torch.randn(co, ci, device=dev) * 0.1
    A = torch.rand(NB, ci, device=dev, dtype=torch.float64) + 0.5
    B = torch.randn(NB, ci, device=dev, dtype=torch.float64) * 0.1
    s = torch.zeros(NB, co, device=dev, dtype=torch.float64); q = torch.zeros_like(s)
    gs = torch.randn(NB, co, device=dev, dtype=torch.float64) * 1e-3; gq = torch.randn_like(gs) * 1e-3
    gA = torch.zeros(NB, ci, device=dev, dtype=torch.float64); gB = torch.zeros_like(gA)
    gw = torch.zeros(co, ci, device=dev, dtype=torch.float64)
    act = 2 if 'conv3' in name else 1
    gb = 2.0 * NB * Q / 1e9
    t_f = timeit(lambda: call('cfn_pwconv_fwd_bf16', x, A, B, act, w, y, s, q, NB, ci, co, Q))
    t_d = timeit(lambda: call('cfn_pwconv_bwd_data_bf16', gy, y, gs, gq, w, x, A, B, act, gx, gA, gB, NB, ci, co, T, H, H, None, 1, None))
    t_d0 = timeit(lambda: call('cfn_pwconv_bwd_data_bf16', gy, y, gs, gq, w, None, None, None, 0, gx, None, None, NB, ci, co, T, H, H, None, 1, None))
    t_d1 = timeit(lambda: call('cfn_pwconv_bwd_data_bf16', gy, None, None, None, w, None, None, None, 0, gx, None, None, NB, ci, co, T, H, H, None, 1, None))
    t_w = timeit(lambda: call('cfn_pwconv_bwd_weight_bf16', gy, y, gs, gq, x, A, B, act, gw, NB, ci, co, Q, None))
    fl = 2.0 * ci * co * NB * Q / 1e12
    print('%-24s fwd %7.3f ms %5.2f TB/s %6.1f TF | dgrad+epi %7.3f ms %5.2f TB/s | dgrad(g,y) %7.3f ms %5.2f TB/s | dgrad(g) %7.3f ms %5.2f TB/s | wgrad %7.3f ms %5.2f TB/s %6.1f TF'
          % (name, t_f, gb * (ci + co) / t_f, fl / t_f * 1e3, t_d, gb * (2 * co + 2 * ci) / t_d, t_d0, gb * (2 * co + ci) / t_d0,
             t_d1, gb * (co + ci) / t_d1, t_w, gb * (2 * co + ci) / t_w, fl / t_w * 1e3), flush=True)
