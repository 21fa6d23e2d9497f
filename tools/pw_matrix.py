#!/usr/bin/env python3
"""Forward-time matrix of one pointwise layer shape over (statistics on/off) x (prologue activation) -- where the time of the
split-bf16 forward kernel goes.  env CFN_PW_SPLIT / CFN_PWS_NP select the kernel.  usage: pw_matrix.py Cin Cout H [T] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import cfn_hip
from cfn_hip import ops
ci, co, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
NB = int(sys.argv[5]) if len(sys.argv) > 5 else 8
x = torch.randn(NB, ci, T, H, H, device='cuda')
w = torch.randn(co, ci, 1, 1, 1, device='cuda') * 0.1
A = torch.rand(NB, ci, device='cuda') + 0.5
B = torch.randn(NB, ci, device='cuda') * 0.1
gb = 4.0 * NB * T * H * H * (ci + co) / 1e9
for stats in (True, False):
    for act in (0, 1, 2):
        def f():
            ops.pwconv(x, w, A, B, act, 1, stats)
        f(); torch.cuda.synchronize()
        cfn_hip.prof_enable('pwconv_fwd', True)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        cfn_hip.prof_enable('pwconv_fwd', False)
        ms, n, _ = cfn_hip.prof_collect('pwconv_fwd')
        print('split=%s np=%s stats=%d act=%d  %.3f ms  %.0f GB/s' % (os.environ.get('CFN_PW_SPLIT', '-'), os.environ.get('CFN_PWS_NP', '-'), stats, act, ms / n, gb / (ms / n) * 1e3))
