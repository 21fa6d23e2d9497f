#!/usr/bin/env python3
"""Random-shape fuzz of the conv operators against fp64 references computed with plain tensor ops (no MIOpen): depthwise 3x3x3 (stride 1 / 2),
pointwise (stride 1 / 2, prologue, statistics), their backward passes.  Every case prints its shape BEFORE it launches (a device fault kills the
process: the last line names the culprit).   python tools/fuzz_ops.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402
import torch.nn.functional as F   # noqa: E402
from cfn_hip import ops           # noqa: E402

DEV = 'cuda'


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def act_ref(z, act):
    return z if act == 0 else (z.clamp(min=0) if act == 1 else z * torch.sigmoid(z))


def dw_case(g, rng):
    N, C, T = rng.choice([1, 2, 3]), rng.choice([1, 3, 8, 24, 54, 108, 216, 432, 7, 33]), rng.choice([1, 2, 3, 5, 8, 9, 16, 17])
    H = rng.choice([1, 2, 3, 5, 7, 8, 13, 14, 15, 28, 29, 56, 57, 112])
    W = rng.choice([H, H, H + 1, max(1, H - 1), 7, 12])
    if C * T * H * W > 4e6:
        T = max(1, int(4e6 // (C * H * W)))
    stride, act, pro = rng.choice([1, 2]), rng.choice([0, 1, 2]), rng.choice([True, False])
    print('dw  N=%d C=%d T=%d H=%d W=%d stride=%d act=%d prologue=%s' % (N, C, T, H, W, stride, act, pro), flush=True)
    x = torch.randn(N, C, T, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(C, 1, 3, 3, 3, generator=g) * 0.2).to(DEV).requires_grad_(True)
    A = (1 + 0.2 * torch.randn(N, C, generator=g)).to(DEV).requires_grad_(True) if pro else None
    B = (0.3 * torch.randn(N, C, generator=g)).to(DEV).requires_grad_(True) if pro else None
    y, s, q = ops.dwconv3d(x, w, A, B, act if pro else 0, stride, True)
    xd = x.detach().double().requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    Ad = A.detach().double().requires_grad_(True) if pro else None
    Bd = B.detach().double().requires_grad_(True) if pro else None
    z = act_ref(xd * Ad.view(N, C, 1, 1, 1) + Bd.view(N, C, 1, 1, 1), act) if pro else xd
    yr = F.conv3d(z, wd, stride=(1, stride, stride), padding=1, groups=C)
    e = [relerr(y, yr), relerr(s, yr.sum((2, 3, 4))), relerr(q, (yr * yr).sum((2, 3, 4)))]
    gy = torch.randn(y.shape, generator=g).to(DEV)
    ins = (x, w) + ((A, B) if pro else ())
    gr = torch.autograd.grad((y,), ins, (gy,))
    grr = torch.autograd.grad((yr,), (xd, wd) + ((Ad, Bd) if pro else ()), (gy.double(),))
    e += [relerr(a, b) for a, b in zip(gr, grr)]
    return max(e), e


def pw_case(g, rng):
    N, Ci, Co = rng.choice([1, 2, 3]), rng.choice([3, 24, 48, 54, 96, 108, 192, 216, 432, 50, 100]), rng.choice([1, 24, 33, 48, 54, 96, 108, 157, 192, 216, 432])
    T, H = rng.choice([1, 2, 3, 4, 5, 8, 13]), rng.choice([1, 2, 3, 4, 7, 8, 14, 15, 28])
    W = rng.choice([H, H, H + 1, 6])
    if max(Ci, Co) * T * H * W > 3e6:
        T = max(1, int(3e6 // (max(Ci, Co) * H * W)))
    stride, act, pro = rng.choice([1, 1, 1, 2]), rng.choice([0, 1, 2]), rng.choice([True, False])
    print('pw  N=%d Cin=%d Cout=%d T=%d H=%d W=%d stride=%d act=%d prologue=%s' % (N, Ci, Co, T, H, W, stride, act, pro), flush=True)
    x = torch.randn(N, Ci, T, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Co, Ci, 1, 1, 1, generator=g) * (2.0 / Ci) ** 0.5).to(DEV).requires_grad_(True)
    A = (1 + 0.2 * torch.randn(N, Ci, generator=g)).to(DEV).requires_grad_(True) if pro else None
    B = (0.3 * torch.randn(N, Ci, generator=g)).to(DEV).requires_grad_(True) if pro else None
    y, s, q = ops.pwconv(x, w, A, B, act if pro else 0, stride, True)
    xd = x.detach().double().requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    Ad = A.detach().double().requires_grad_(True) if pro else None
    Bd = B.detach().double().requires_grad_(True) if pro else None
    z = act_ref(xd * Ad.view(N, Ci, 1, 1, 1) + Bd.view(N, Ci, 1, 1, 1), act) if pro else xd
    yr = torch.einsum('nkthw,mk->nmthw', z[:, :, :, ::stride, ::stride], wd.view(Co, Ci))
    e = [relerr(y, yr), relerr(s, yr.sum((2, 3, 4))), relerr(q, (yr * yr).sum((2, 3, 4)))]
    gy = torch.randn(y.shape, generator=g).to(DEV)
    gs, gq = (0.01 * torch.randn(s.shape, generator=g)).to(DEV).to(s.dtype), (0.001 * torch.randn(q.shape, generator=g)).to(DEV).to(q.dtype)
    ins = (x, w) + ((A, B) if pro else ())
    gr = torch.autograd.grad((y, s, q), ins, (gy, gs, gq))
    sr, qr = yr.sum((2, 3, 4)), (yr * yr).sum((2, 3, 4))
    grr = torch.autograd.grad((yr, sr, qr), (xd, wd) + ((Ad, Bd) if pro else ()), (gy.double(), gs.double(), gq.double()))
    e += [relerr(a, b) for a, b in zip(gr, grr)]
    return max(e), e


def t5_case(g, rng):
    N, C, T = rng.choice([1, 2]), rng.choice([1, 8, 24, 25]), rng.choice([1, 2, 4, 5, 9, 16, 33])
    H, W = rng.choice([1, 3, 8, 28, 56, 112]), rng.choice([1, 3, 8, 28, 56, 112, 5])
    if C * T * H * W > 3e6:
        T = max(1, int(3e6 // (C * H * W)))
    print('t5  N=%d C=%d T=%d H=%d W=%d' % (N, C, T, H, W), flush=True)
    x = torch.randn(N, C, T, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(C, 1, 5, 1, 1, generator=g) * 0.3).to(DEV).requires_grad_(True)
    y, s, q = ops.dwconv_t5(x, w, True)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = F.conv3d(xd, wd, padding=(2, 0, 0), groups=C)
    e = [relerr(y, yr), relerr(s, yr.sum((2, 3, 4))), relerr(q, (yr * yr).sum((2, 3, 4)))]
    gy = torch.randn(y.shape, generator=g).to(DEV)
    gr = torch.autograd.grad((y,), (x, w), (gy,))
    grr = torch.autograd.grad((yr,), (xd, wd), (gy.double(),))
    e += [relerr(a, b) for a, b in zip(gr, grr)]
    return max(e), e


def tail_case(g, rng):
    N, C, T, H = rng.choice([1, 2, 3]), rng.choice([1, 24, 48, 96, 192, 7]), rng.choice([1, 2, 5, 8]), rng.choice([1, 3, 7, 14, 28, 56, 9])
    W = rng.choice([H, H + 1, 4])
    two = rng.choice([True, False])
    print('tail N=%d C=%d T=%d H=%d W=%d shortcut-bn=%s' % (N, C, T, H, W, two), flush=True)
    mk = lambda sc, off=0.0: (off + sc * torch.randn(N, C, generator=g)).to(DEV).requires_grad_(True)
    y = torch.randn(N, C, T, H, W, generator=g).to(DEV).requires_grad_(True)
    res = torch.randn(N, C, T, H, W, generator=g).to(DEV).requires_grad_(True)
    A, B = mk(0.2, 1.0), mk(0.3)
    Ar, Br = (mk(0.2, 1.0), mk(0.3)) if two else (None, None)
    out = ops.bn_add_relu(y, A, B, res, Ar, Br)
    d = lambda t: None if t is None else t.detach().double().requires_grad_(True)
    yd, rd, Ad, Bd, Ard, Brd = d(y), d(res), d(A), d(B), d(Ar), d(Br)
    v = lambda c: c.view(N, C, 1, 1, 1)
    ref = (yd * v(Ad) + v(Bd) + (rd * v(Ard) + v(Brd) if two else rd)).clamp(min=0)
    e = [relerr(out, ref)]
    go = torch.randn(out.shape, generator=g).to(DEV)
    ins = (y, A, B, res) + ((Ar, Br) if two else ())
    gr = torch.autograd.grad((out,), ins, (go,))
    grr = torch.autograd.grad((ref,), (yd, Ad, Bd, rd) + ((Ard, Brd) if two else ()), (go.double(),))
    e += [relerr(a, b) for a, b in zip(gr, grr)]
    return max(e), e


def pool_case(g, rng):
    N, C, T, H = rng.choice([1, 2]), rng.choice([1, 24, 192, 432, 5]), rng.choice([1, 3, 8]), rng.choice([7, 14, 28, 56, 8, 12])
    W = H
    O = rng.choice([o for o in (1, 2, 7, H) if H % o == 0])
    act, pro = rng.choice([0, 1, 2]), rng.choice([True, False])
    print('pool N=%d C=%d T=%d H=%d -> %d act=%d prologue=%s' % (N, C, T, H, O, act, pro), flush=True)
    x = torch.randn(N, C, T, H, W, generator=g).to(DEV).requires_grad_(True)
    A = (1 + 0.2 * torch.randn(N, C, generator=g)).to(DEV).requires_grad_(True) if pro else None
    B = (0.3 * torch.randn(N, C, generator=g)).to(DEV).requires_grad_(True) if pro else None
    out = ops.pool_hw(x, O, O, A, B, act if pro else 0)
    xd = x.detach().double().requires_grad_(True)
    Ad = A.detach().double().requires_grad_(True) if pro else None
    Bd = B.detach().double().requires_grad_(True) if pro else None
    z = act_ref(xd * Ad.view(N, C, 1, 1, 1) + Bd.view(N, C, 1, 1, 1), act) if pro else xd
    ref = F.adaptive_avg_pool3d(z, (T, O, O))
    e = [relerr(out, ref)]
    go = torch.randn(out.shape, generator=g).to(DEV)
    gr = torch.autograd.grad((out,), (x,) + ((A, B) if pro else ()), (go,))
    grr = torch.autograd.grad((ref,), (xd,) + ((Ad, Bd) if pro else ()), (go.double(),))
    e += [relerr(a, b) for a, b in zip(gr, grr)]
    return max(e), e


def dense_case(g, rng):
    """dense implicit-GEMM conv of the Grid Pool saliency branch (csrc/salconv.hip, salconvb.hip, the im2col route): fp64 reference on the CPU"""
    N, Ci, Co = rng.choice([1, 2]), rng.choice([3, 4, 5, 8, 24, 24, 24]), rng.choice([1, 4, 7, 20, 24, 24, 40])
    T, H = rng.choice([1, 2, 5, 6, 9, 12, 17]), rng.choice([7, 8, 9, 14, 16, 20, 28, 56])
    W = rng.choice([H, H, 28, 56, 16, 7])
    k, st, pd = rng.choice([((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((1, 3, 3), (1, 2, 2), (0, 1, 1))])
    act, pro = rng.choice([0, 1]), rng.choice([True, False])      # (the op refuses a swish prologue: the branch has ReLUs only)
    print('dense N=%d Ci=%d Co=%d T=%d H=%d W=%d k=%s act=%d prologue=%s' % (N, Ci, Co, T, H, W, k, act, pro), flush=True)
    x = torch.randn(N, Ci, T, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(Co, Ci, *k, generator=g) * (2.0 / (Ci * k[0] * 9)) ** 0.5).to(DEV).requires_grad_(True)
    A = (1 + 0.2 * torch.randn(N, Ci, generator=g)).to(DEV).requires_grad_(True) if pro else None
    B = (0.3 * torch.randn(N, Ci, generator=g)).to(DEV).requires_grad_(True) if pro else None
    y, s, q = ops.conv3d_dense(x, w, k, st, pd, A, B, act if pro else 0, True)
    xd, wd = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
    Ad = A.detach().double().cpu().requires_grad_(True) if pro else None
    Bd = B.detach().double().cpu().requires_grad_(True) if pro else None
    z = act_ref(xd * Ad.view(N, Ci, 1, 1, 1) + Bd.view(N, Ci, 1, 1, 1), act) if pro else xd
    ref = F.conv3d(z, wd, stride=st, padding=pd)
    e = [relerr(y.cpu(), ref), relerr(s.cpu(), ref.sum((2, 3, 4))), relerr(q.cpu(), (ref * ref).sum((2, 3, 4)))]
    go = torch.randn(y.shape, generator=g)
    gs, gq = torch.randn(s.shape, generator=g) * 0.01, torch.randn(q.shape, generator=g) * 0.001
    ins = (x, w) + ((A, B) if pro else ())
    gr = torch.autograd.grad((y, s, q), ins, (go.to(DEV), gs.to(DEV).to(s.dtype), gq.to(DEV).to(q.dtype)))
    loss = (ref * go.double()).sum() + (ref.sum((2, 3, 4)) * gs.double()).sum() + ((ref * ref).sum((2, 3, 4)) * gq.double()).sum()
    grr = torch.autograd.grad(loss, (xd, wd) + ((Ad, Bd) if pro else ()))
    e += [relerr(a.cpu(), b) for a, b in zip(gr, grr)]
    return max(e), e


def gather_case(g, rng):
    """temporal-alignment gather of the fusion branch (csrc/fusion.hip, register-tiled kernels) against the fp64 expression of x3d_coarse.py:209-223"""
    B, crops, C = rng.choice([1, 2, 3]), rng.choice([1, 1, 2, 3]), rng.choice([5, 8, 24, 48, 96, 192, 432])
    Tf, K, P = rng.choice([7, 12, 40, 65, 128]), rng.choice([3, 5, 17, 33, 65]), rng.choice([1, 49, 49])
    print('gather B=%d crops=%d C=%d Tf=%d K=%d P=%d' % (B, crops, C, Tf, K, P), flush=True)
    x = torch.randn(B, C, Tf, P, generator=g).abs().to(DEV).requires_grad_(True)
    at = torch.randn(B, Tf, P, generator=g).to(DEV).requires_grad_(True)
    bias = torch.tensor([0.3]).to(DEV).requires_grad_(True)
    GX = torch.randn(B * crops, Tf, K, generator=g).abs().to(DEV).requires_grad_(True)
    mask = torch.ones(B, Tf)
    mask[:, -rng.choice([1, 2, 3]):] = 0
    mask = mask.to(DEV)
    z = ops.fusion_gather(x, at, bias, GX, mask, crops)
    d = [v.detach().double().requires_grad_(True) for v in (x, at, bias, GX)]
    rep = lambda v: v.unsqueeze(1).repeat((1, crops) + (1,) * (v.dim() - 1)).view((B * crops,) + tuple(v.shape[1:]))
    a2 = rep(torch.sigmoid(d[1] + d[2]))
    wgt = a2.unsqueeze(2) * (d[3] * rep(mask.double()).unsqueeze(2)).unsqueeze(3)
    ref = torch.einsum('bctp,btkp->bckp', rep(d[0]), wgt) / (wgt.sum(1) + 1e-6).unsqueeze(1)
    e = [relerr(z, ref)]
    go = torch.randn(z.shape, generator=g).to(DEV)
    gr = torch.autograd.grad(z, (x, at, bias, GX), go)
    grr = torch.autograd.grad(ref, d, go.double())
    e += [relerr(a_, b_) for a_, b_ in zip(gr, grr)]
    # the attention bias is ONE scalar: its gradient is an fp32 sum over B x Tf x P signed terms (as in the reference) and carries the cancellation
    # of that sum (1e-3 relative at Tf = 128 in the first campaign); it is judged against 5e-3, everything else against the common bound
    return max(e[:3] + [e[3] * 0.04] + e[4:]), e


def main():
    import random
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--all', action='store_true', help='also conv1_t (5x1x1 depthwise), the block tail, the spatial pooling, the dense saliency conv and the fusion gather')
    args = ap.parse_args()
    rng = random.Random(args.seed)
    g = torch.Generator().manual_seed(args.seed)
    bad = 0
    for i in range(args.cases):
        fn = (dw_case, pw_case, t5_case, tail_case, pool_case, dense_case, gather_case)[i % 7] if args.all else (dw_case if i % 2 == 0 else pw_case)
        try:
            worst, e = fn(g, rng)
        except RuntimeError as ex:
            print('   -> refused / error: %s' % str(ex)[:160], flush=True)
            continue
        torch.cuda.synchronize()
        if not worst <= 2e-4:
            bad += 1
            print('   -> MISMATCH %s' % ['%.1e' % v for v in e], flush=True)
    print('fuzz: %d cases, %d mismatches' % (args.cases, bad))


if __name__ == '__main__':
    main()
