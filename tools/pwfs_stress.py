#!/usr/bin/env python3
"""Bit-repeat stress of the one-pass split-bf16 pointwise backward (csrc/pwfuseds.hip): R launches on the same inputs, every output compared
bit for bit with the first launch's; reports which output differs, in how many elements and where.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402

DEV = 'cuda'
os.environ['CFN_PWF_SPLIT'] = '2'
R = int(os.environ.get('RUNS', '200'))
CFGS = [(1, 128, 64, 4, 28, 28, 1), (1, 108, 48, 3, 10, 10, 1), (2, 48, 108, 3, 8, 8, 1), (2, 108, 48, 8, 28, 28, 2), (2, 48, 108, 8, 28, 28, None),
        (2, 108, 48, 8, 28, 28, 0), (4, 48, 108, 16, 28, 28, 1), (2, 24, 108, 8, 56, 56, None), (2, 96, 216, 16, 14, 14, None), (8, 96, 216, 8, 14, 14, None), (2, 48, 216, 8, 28, 28, None)]
for N, Cin, Cout, T, H, W, act in CFGS:
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g)
    gy, y, x = rnd(N, Cout, T, H, W).to(DEV), rnd(N, Cout, T, H, W).to(DEV), rnd(N, Cin, T, H, W).to(DEV)
    w = (0.3 * rnd(Cout, Cin)).to(DEV)
    f64 = lambda *s, scale=1.0: (rnd(*s) * scale).double().to(DEV)
    gs, gq, gsc = f64(N, Cout, scale=0.05), f64(N, Cout, scale=0.01), 1.0 + f64(N, Cout, scale=0.3)
    A = B = None
    if act is not None:
        A, B = 1.0 + f64(N, Cin, scale=0.2), f64(N, Cin, scale=0.2)

    def run():
        gx = torch.full_like(x, float('nan'))
        gA = gB = None
        if A is not None:
            gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        assert cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, A, B, act or 0, gx, gA, gB, gw, N, Cin, Cout, T, H, W, None, 1, gsc)
        return {'gx': gx, 'gA': gA, 'gB': gB, 'gw': gw}

    ref = run()
    bad = {}
    for r in range(R):
        out = run()
        for k, v in out.items():
            if v is not None and not torch.equal(v, ref[k]):
                d = (v != ref[k]) & ~(torch.isnan(v) & torch.isnan(ref[k]))
                idx = d.nonzero()
                bad.setdefault(k, []).append((r, int(d.sum()), idx[:6].tolist(), [float(v[tuple(i)]) for i in idx[:3]], [float(ref[k][tuple(i)]) for i in idx[:3]]))
    print('N=%d %d->%d T=%d %dx%d act=%s: %d launches; differing outputs: %s' % (N, Cin, Cout, T, H, W, act, R, {k: len(v) for k, v in bad.items()} or 'none'), flush=True)
    for k, v in bad.items():
        for item in v[:4]:
            print('   ', k, item)
