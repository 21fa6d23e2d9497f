"""GPU parity of the split-bf16 pointwise kernels (csrc/pwsplit.hip, pwsplitw.hip; conv1x1x1 of the reference,
x3d_fine.py:100-105): fp32 tensors, operands split into 2 / 3 bf16 terms, 3 / 6 bf16 MFMAs per k-block.

Checked against fp32 torch on the CPU (same tolerances as the fp32-MFMA kernels: forward 3e-5, gradients 3e-4 of the
largest reference value), against an fp64 reference (the 6-term product must be as accurate as the fp32-MFMA kernel,
the 3-term product within 2e-5), and -- through the C ABI -- against the fp32-MFMA kernels on the same device buffers
for everything the backward entry points take (statistics gradients, tail scale, compact shortcut gradient)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr
from test_hip_ops import DEV, check_conv, ops, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture
def split():
    import cfn_hip
    prev = cfn_hip.query('cfn_pw_split_terms', -1)

    def set_terms(t):
        cfn_hip.query('cfn_pw_split_terms', t)
    yield set_terms
    cfn_hip.query('cfn_pw_split_terms', prev)


def test_split_terms_setting():
    import cfn_hip
    prev = cfn_hip.query('cfn_pw_split_terms', -1)
    assert prev in (0, 3, 6)
    assert cfn_hip.query('cfn_pw_split_terms', 3) == prev
    assert cfn_hip.query('cfn_pw_split_terms', -1) == 3
    assert cfn_hip.query('cfn_pw_split_terms', 5) == -2 and cfn_hip.query('cfn_pw_split_terms', -1) == 3
    cfn_hip.query('cfn_pw_split_terms', prev)


# shapes inside the split kernels' range (Cin >= 48, Cout > 32 [weight gradient: both >= 48], even position count)
SPLIT_CASES = [
    # N, Cin, Cout, T, H, W, act, pro
    (2, 48, 108, 2, 6, 6, 1, True),        # 1 slab of 4 row tiles (ragged: 108 rows)
    (1, 108, 48, 3, 6, 5, 2, True),        # K not a multiple of 16 / 32, 90 positions (ragged last tile)
    (1, 96, 216, 2, 7, 8, 1, True),        # 2 slabs forward
    (1, 216, 96, 2, 14, 14, 2, True),      # X3D layer-3 conv3, several position tiles per wave
    (1, 192, 432, 3, 4, 4, 0, False),      # 4 slabs, no prologue
    (1, 432, 192, 2, 4, 3, 2, True),       # deep K: slabs limited by LDS
    (2, 50, 70, 1, 5, 6, 2, True),         # nothing aligned: K % 16 = 2, M % 32 = 6, 30 positions
    (1, 64, 33, 2, 2, 2, 0, True),         # M = 33: two row tiles, one of a single row (forward / data gradient only)
    (1, 432, 2048, 4, 1, 1, 0, False),     # fc1-like: 4 positions
]


@pytest.mark.parametrize('terms', [3, 6])
@pytest.mark.parametrize('N,Cin,Cout,T,H,W,act,pro', SPLIT_CASES)
def test_pwconv_split(split, terms, N, Cin, Cout, T, H, W, act, pro):
    split(terms)
    x, w = rnd(1, N, Cin, T, H, W), rnd(2, Cout, Cin, 1, 1, 1, scale=(2.0 / Cin) ** 0.5)
    A = (1 + 0.2 * rnd(3, N, Cin)) if pro else None
    B = 0.3 * rnd(4, N, Cin) if pro else None
    check_conv(lambda x_, w_, A_, B_: ops().pwconv(x_, w_, A_, B_, act, 1, True),
               lambda a, w_: F.conv3d(a, w_), x, w, A, B, act, tol_f=3e-5, tol_g=3e-4)


def _ref64(x, w, A, B, act):
    """fp64 forward / backward of act(A x + B) -> 1x1x1 conv with a random cotangent; returns (y, gx, gw)"""
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    shp = (x.shape[0], -1, 1, 1, 1)
    z = xd * A.double().view(shp) + B.double().view(shp)
    a = F.relu(z) if act == 1 else (z * torch.sigmoid(z) if act == 2 else z)
    y = F.conv3d(a, wd)
    r = rnd(7, *y.shape).double()
    gx, gw = torch.autograd.grad((y * r).sum(), (xd, wd))
    return y, gx, gw, r


@pytest.mark.parametrize('cfg', [(1, 96, 216, 4, 14, 14, 1), (1, 216, 96, 4, 14, 14, 2), (1, 432, 192, 6, 7, 7, 2)])
def test_split_accuracy_vs_fp64(split, cfg):
    """errors against an fp64 reference, fp32-MFMA kernel next to the 3- and 6-term split: 6 terms is an fp32 product
    (dropped terms 3*2^-27), 3 terms drops 3*2^-18 per product"""
    N, Cin, Cout, T, H, W, act = cfg
    x, w = rnd(1, N, Cin, T, H, W), rnd(2, Cout, Cin, 1, 1, 1, scale=(2.0 / Cin) ** 0.5)
    A, B = 1 + 0.2 * rnd(3, N, Cin), 0.3 * rnd(4, N, Cin)
    y64, gx64, gw64, r = _ref64(x, w, A, B, act)
    err = {}
    for terms in (0, 3, 6):
        split(terms)
        xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        y, _, _ = ops().pwconv(xg, wg, A.to(DEV), B.to(DEV), act, 1, True)
        gx, gw = torch.autograd.grad((y * r.float().to(DEV)).sum(), (xg, wg))
        err[terms] = (relerr(y, y64), relerr(gx, gx64), relerr(gw, gw64))
    print('split accuracy vs fp64 (y, gx, gw):', cfg, {k: ['%.2e' % e for e in v] for k, v in err.items()})
    for i in range(3):
        assert err[6][i] <= 2.0 * err[0][i] + 2e-7, (i, err)
        assert err[3][i] <= 2e-5, (i, err)


@pytest.mark.parametrize('terms', [3, 6])
@pytest.mark.parametrize('act', [None, 0, 1, 2])
@pytest.mark.parametrize('cfg', [(2, 48, 108, 3, 8, 8, 0), (1, 108, 48, 2, 12, 12, 2), (1, 96, 216, 2, 7, 7, 2), (1, 216, 96, 3, 6, 6, 0),
                                 (1, 60, 50, 2, 6, 5, 2), (1, 192, 432, 2, 4, 4, 0)])
def test_split_backward_matches_fp32_mfma(split, cfg, act, terms):
    """cfn_pwconv_bwd_data_acc / cfn_pwconv_bwd_weight with the split switched on against the fp32-MFMA kernels on the same
    buffers: with / without prologue (act None = no A, B), statistics gradients, the tail scale `gscale`, the compact
    shortcut gradient `acc` on its stride lattice (odd width: both positions of a pair can be on it)."""
    import cfn_hip
    N, Cin, Cout, T, H, W, acc_s = cfg
    f64 = lambda seed, *shape, scale=1.0: (rnd(seed, *shape) * scale).double().to(DEV)
    gy, y, x = rnd(1, N, Cout, T, H, W).to(DEV), rnd(2, N, Cout, T, H, W).to(DEV), rnd(3, N, Cin, T, H, W).to(DEV)
    w = (0.3 * rnd(4, Cout, Cin)).to(DEV)
    gs, gq, gsc = f64(5, N, Cout, scale=0.05), f64(6, N, Cout, scale=0.01), 1.0 + f64(7, N, Cout, scale=0.3)
    A = B = None
    if act is not None:
        A, B = 1.0 + f64(8, N, Cin, scale=0.2), f64(9, N, Cin, scale=0.2)
    acc = None
    if acc_s:
        acc = rnd(10, N, Cin, T, (H - 1) // acc_s + 1, (W - 1) // acc_s + 1).to(DEV)

    def run(t):
        split(t)
        gx = torch.empty_like(x)
        gA = gB = None
        if A is not None:
            gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        a_ = 0 if act is None else act
        cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, N, Cin, Cout, T, H, W, 1, acc,
                     acc_s or 1, gsc)
        cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, a_, gw, N, Cin, Cout, T, H, W, 1, gsc)
        return gx, gA, gB, gw

    ref, got = run(0), run(terms)
    for name, r, g in zip(('gx', 'gA', 'gB', 'gw'), ref, got):
        if r is not None:
            assert relerr(g, r) <= (2e-5 if terms == 3 else 3e-6), (name, relerr(g, r))


@pytest.mark.parametrize('terms', [3, 6])
def test_split_bitwise_reproducible(split, terms):
    split(terms)
    o = ops()
    x, w = rnd(1, 2, 96, 4, 14, 14).to(DEV), (0.2 * rnd(2, 216, 96, 1, 1, 1)).to(DEV)
    A, B = (1 + 0.2 * rnd(3, 2, 96)).to(DEV), (0.1 * rnd(4, 2, 96)).to(DEV)
    runs = []
    for _ in range(3):
        leaves = [v.clone().requires_grad_(True) for v in (x, w, A, B)]
        y, s, q = o.pwconv(*leaves, 2, 1, True)
        ((y * y).sum() + s.sum() + 0.1 * q.sum()).backward()
        runs.append([y.detach(), s.detach(), q.detach()] + [v.grad for v in leaves])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)


# ---- register-resident-weights kernel (csrc/pwregk.hip): 128 < Cin (contraction) <= 224 with an even number of 16-channel blocks,
# 32 < rows <= 128, position count a multiple of 4; forward and the data gradient without act' epilogue
PWK_CASES = [
    # N, K (contraction), M (rows), T, H, W
    (2, 216, 96, 2, 14, 14),      # X3D layer-3 conv3 forward / conv1 data gradient
    (1, 216, 96, 5, 14, 14),      # 30.6 position tiles: ragged last tile, several tiles per workgroup
    (3, 216, 96, 1, 14, 14),      # one frame
    (2, 160, 64, 3, 6, 6),        # 10 k-blocks, two row tiles
    (1, 224, 128, 2, 4, 4),       # the largest shape served: 14 full k-blocks, 4 full row tiles, ONE position tile
    (1, 192, 33, 2, 4, 4),        # 33 rows: a row tile of a single row
    (1, 210, 100, 3, 10, 10),     # K % 16 = 2 (zero-padded contraction), 100 rows (ragged row tile), 300 positions
]


# one-slice mode (forward only): K <= 112, 128 < rows <= 256 -- wave = row tile, the whole contraction in its registers
PWK1_CASES = [
    (2, 96, 216, 2, 14, 14),      # X3D layer-3 conv1 forward
    (1, 96, 216, 5, 14, 14),
    (2, 48, 160, 3, 6, 6),        # 3 k-blocks, 5 row tiles
    (1, 80, 250, 3, 10, 10),      # 250 rows: 8 row tiles, the last ragged
    (1, 112, 256, 2, 4, 4),       # the largest shape served
    (1, 64, 129, 2, 4, 4),        # 129 rows: a row tile of a single row
    (1, 100, 200, 1, 6, 6),       # K % 16 = 4
    (2, 192, 432, 2, 7, 7),       # X3D layer-4 conv1 forward: two slabs of 7 row tiles, 12 k-blocks in registers, 98 positions (not % 4: declined)
    (1, 192, 432, 4, 7, 7),       # the same with 196 positions
    (1, 176, 300, 2, 4, 4),       # 11 k-blocks: not instantiated, falls through
    (1, 192, 512, 1, 4, 4),       # 16 row tiles: two slabs of 8
]


@pytest.mark.parametrize('N,K,M,T,H,W', PWK_CASES + PWK1_CASES)
@pytest.mark.parametrize('act', [0, 2])
def test_pwk_forward_vs_fp64(split, N, K, M, T, H, W, act):
    """forward with prologue + statistics against fp64: an fp32-accurate product (6-term split), sums within fp32 rounding of the
    fp64 sums, and the same bits on a second run (no atomics on y)"""
    split(6)
    x, w = rnd(1, N, K, T, H, W).to(DEV), rnd(2, M, K, 1, 1, 1, scale=(2.0 / K) ** 0.5).to(DEV)
    A, B = (1 + 0.2 * rnd(3, N, K)).to(DEV), (0.3 * rnd(4, N, K)).to(DEV)
    y, s, q = ops().pwconv(x, w, A, B, act, 1, True)
    y2, _, _ = ops().pwconv(x, w, A, B, act, 1, True)
    assert torch.equal(y, y2)
    z = x.double() * A.double().view(N, K, 1, 1, 1) + B.double().view(N, K, 1, 1, 1)
    z = z * torch.sigmoid(z) if act == 2 else z
    yr = torch.einsum('nkthw,mk->nmthw', z, w.double().view(M, K))
    assert relerr(y.double(), yr) <= 1e-6
    assert relerr(s, yr.sum((2, 3, 4))) <= 2e-6 and relerr(q, (yr * yr).sum((2, 3, 4))) <= 2e-6


@pytest.mark.parametrize('N,K,M,T,H,W', PWK_CASES)
@pytest.mark.parametrize('two', [True, False])
def test_pwk_data_gradient_vs_fp64(split, N, K, M, T, H, W, two):
    """data gradient of a prologue-free conv M -> K channels (contraction over its K output channels, statistics gradients folded into
    the staged operand g' = gy + gs + 2 gq y; `two`: with the y term) against fp64, bit-repeatable"""
    split(6)
    x = rnd(1, N, M, T, H, W).to(DEV).requires_grad_(True)
    w = rnd(2, K, M, 1, 1, 1, scale=(2.0 / M) ** 0.5).to(DEV).requires_grad_(True)
    y, s, q = ops().pwconv(x, w, None, None, 0, 1, True)
    gy, gs, gq = rnd(5, *y.shape).to(DEV), (0.01 * rnd(6, *s.shape)).to(DEV).to(s.dtype), (0.001 * rnd(7, *q.shape)).to(DEV).to(q.dtype)
    outs, gos = ((y, s, q), (gy, gs, gq)) if two else ((y, s), (gy, gs))
    gx, = torch.autograd.grad(outs, (x,), gos, retain_graph=True)
    gx2, = torch.autograd.grad(outs, (x,), gos, retain_graph=True)
    assert torch.equal(gx, gx2)
    wd = w.detach().double().view(K, M)
    yd = torch.einsum('nmthw,km->nkthw', x.detach().double(), wd)
    gp = gy.double() + gs.double().view(N, K, 1, 1, 1) + (2.0 * yd * gq.double().view(N, K, 1, 1, 1) if two else 0.0)
    gxr = torch.einsum('nkthw,km->nmthw', gp, wd)
    assert relerr(gx.double(), gxr) <= 1e-6


# pws_kernel with STREAMED weights (csrc/pwstream.hip, round 5): the deep contractions of layer 4 (K >= 400): weights pre-split into a
# workspace, two LDS chunk buffers of 3 k-blocks, slabs of up to 192 rows, one barrier per chunk, a uniform number of tile rounds per workgroup
PWT_CASES = [
    (2, 432, 192, 4, 7, 7),       # X3D layer-4 conv3 forward / conv1 data gradient: one slab of 6 row tiles, 196 positions (6.1 tiles: ONE round, a wave without a tile)
    (1, 432, 192, 16, 7, 7),      # 24.5 tiles on 4 workgroups
    (8, 432, 192, 40, 7, 7),      # 8 clips, 61.25 tiles per clip on 8 workgroups of 8 waves: ONE full round and a ragged one (odd chunk count: the buffer parity flips)
    (1, 432, 96, 4, 14, 14),      # layer-4 block-0 conv1 data gradient: 3 row tiles
    (1, 432, 432, 4, 7, 7),       # conv5 data gradient: three slabs of 5 row tiles, the last with 112 rows (ragged)
    (1, 432, 33, 2, 4, 4),        # 33 rows, ONE position tile
    (2, 448, 100, 1, 6, 6),       # K padded to 480 (10 chunks), 100 rows, 36 positions
    (1, 416, 64, 3, 4, 4),        # 26 k-blocks: a padding k-block in the last chunk; 64 rows run as 3 row tiles
]


@pytest.mark.parametrize('N,K,M,T,H,W', PWT_CASES)
@pytest.mark.parametrize('act', [0, 2])
def test_pwt_forward_vs_fp64(split, N, K, M, T, H, W, act):
    """the streamed-weight kernel forward with prologue + statistics against fp64 (fp32-accurate 6-term product), bit-repeatable, and different in the
    low bits from the fp32-MFMA kernel the same call used before (i.e. it is the kernel that runs)"""
    import cfn_hip
    split(6)
    x, w = rnd(1, N, K, T, H, W).to(DEV), rnd(2, M, K, 1, 1, 1, scale=(2.0 / K) ** 0.5).to(DEV)
    A, B = (1 + 0.2 * rnd(3, N, K)).to(DEV), (0.3 * rnd(4, N, K)).to(DEV)
    y, s, q = ops().pwconv(x, w, A, B, act, 1, True)
    for _ in range(5):
        y2, s2, q2 = ops().pwconv(x, w, A, B, act, 1, True)
        assert torch.equal(y, y2)
    z = x.double() * A.double().view(N, K, 1, 1, 1) + B.double().view(N, K, 1, 1, 1)
    z = z * torch.sigmoid(z) if act == 2 else z
    yr = torch.einsum('nkthw,mk->nmthw', z, w.double().view(M, K))
    assert relerr(y.double(), yr) <= 1e-6
    assert relerr(s, yr.sum((2, 3, 4))) <= 2e-6 and relerr(q, (yr * yr).sum((2, 3, 4))) <= 2e-6
    y_nostat = ops().pwconv(x, w, A, B, act, 1, False)
    y_nostat = y_nostat[0] if isinstance(y_nostat, (tuple, list)) else y_nostat
    assert torch.equal(y_nostat, y)
    split(0)
    y0, _, _ = ops().pwconv(x, w, A, B, act, 1, True)
    assert relerr(y, y0) <= 3e-6 and not torch.equal(y, y0)


@pytest.mark.parametrize('N,K,M,T,H,W', PWT_CASES)
@pytest.mark.parametrize('two', [True, False])
def test_pwt_data_gradient_vs_fp64(split, N, K, M, T, H, W, two):
    """the streamed-weight kernel data gradient of a prologue-free conv M -> K channels (contraction over its K output channels, g' = gy + gs + 2 gq y)"""
    split(6)
    x = rnd(1, N, M, T, H, W).to(DEV).requires_grad_(True)
    w = rnd(2, K, M, 1, 1, 1, scale=(2.0 / M) ** 0.5).to(DEV).requires_grad_(True)
    y, s, q = ops().pwconv(x, w, None, None, 0, 1, True)
    gy, gs, gq = rnd(5, *y.shape).to(DEV), (0.01 * rnd(6, *s.shape)).to(DEV).to(s.dtype), (0.001 * rnd(7, *q.shape)).to(DEV).to(q.dtype)
    outs, gos = ((y, s, q), (gy, gs, gq)) if two else ((y, s), (gy, gs))
    gx, = torch.autograd.grad(outs, (x,), gos, retain_graph=True)
    for _ in range(5):
        gx2, = torch.autograd.grad(outs, (x,), gos, retain_graph=True)
        assert torch.equal(gx, gx2)
    wd = w.detach().double().view(K, M)
    yd = torch.einsum('nmthw,km->nkthw', x.detach().double(), wd)
    gp = gy.double() + gs.double().view(N, K, 1, 1, 1) + (2.0 * yd * gq.double().view(N, K, 1, 1, 1) if two else 0.0)
    gxr = torch.einsum('nkthw,km->nmthw', gp, wd)
    assert relerr(gx.double(), gxr) <= 1e-6


@pytest.mark.parametrize('N,K,M,T,H,W', [(2, 432, 192, 65, 7, 7), (1, 432, 96, 5, 7, 9), (1, 192, 432, 65, 7, 7)])
@pytest.mark.parametrize('two', [True, False])
def test_pwt_data_gradient_odd_volume(split, N, K, M, T, H, W, two):
    """the coarse stream's layer 4: 65 x 7 x 7 = 3,185 positions per row (rows on 4-byte boundaries only) -- the data gradients (4-byte accesses both
    ways) run on the streamed-weight kernel, the last case with the act' epilogue (contraction over 192, 432 rows); against fp64, bit-repeatable"""
    split(6)
    epi = K < 400
    x = rnd(1, N, M, T, H, W).to(DEV).requires_grad_(True)
    w = rnd(2, K, M, 1, 1, 1, scale=(2.0 / M) ** 0.5).to(DEV)
    A = (1 + 0.2 * rnd(3, N, M)).to(DEV) if epi else None
    B = (0.3 * rnd(4, N, M)).to(DEV) if epi else None
    y, s, q = ops().pwconv(x, w, A, B, 2 if epi else 0, 1, True)
    gy, gs, gq = rnd(5, *y.shape).to(DEV), (0.01 * rnd(6, *s.shape)).to(DEV).to(s.dtype), (0.001 * rnd(7, *q.shape)).to(DEV).to(q.dtype)
    outs, gos = ((y, s, q), (gy, gs, gq)) if two else ((y, s), (gy, gs))
    gx, = torch.autograd.grad(outs, (x,), gos, retain_graph=True)
    for _ in range(3):
        gx2, = torch.autograd.grad(outs, (x,), gos, retain_graph=True)
        assert torch.equal(gx, gx2)
    wd = w.double().view(K, M)
    xd = x.detach().double()
    if epi:
        z = xd * A.double().view(N, M, 1, 1, 1) + B.double().view(N, M, 1, 1, 1)
        sg = torch.sigmoid(z)
        a_, da = z * sg, sg * (1 + z * (1 - sg))
    else:
        a_, da = xd, None
    yd = torch.einsum('nmthw,km->nkthw', a_, wd)
    gp = gy.double() + gs.double().view(N, K, 1, 1, 1) + (2.0 * yd * gq.double().view(N, K, 1, 1, 1) if two else 0.0)
    gxr = torch.einsum('nkthw,km->nmthw', gp, wd)
    if epi:
        gxr = gxr * da * A.double().view(N, M, 1, 1, 1)
    assert relerr(gx.double(), gxr) <= 2e-6


def test_pwt_stress_bit_repeatable(split):
    """200 launches of the layer-4 forward and data gradient between other kernels: every result bit-identical (LDS chunk buffers are overwritten behind ONE barrier per chunk while
    other waves still multiply the previous chunk -- exactly the kind of schedule a missing wait shows up in)"""
    split(6)
    N, K, M, T, H, W = 4, 432, 192, 16, 7, 7
    x, w = rnd(1, N, K, T, H, W).to(DEV), rnd(2, M, K, 1, 1, 1, scale=(2.0 / K) ** 0.5).to(DEV)
    A, B = (1 + 0.2 * rnd(3, N, K)).to(DEV), (0.3 * rnd(4, N, K)).to(DEV)
    xi = rnd(8, N, M, T, H, W).to(DEV).requires_grad_(True)
    wi = rnd(9, K, M, 1, 1, 1, scale=(2.0 / M) ** 0.5).to(DEV)
    yi, si, qi = ops().pwconv(xi, wi, None, None, 0, 1, True)
    gy, gs, gq = rnd(5, *yi.shape).to(DEV), (0.01 * rnd(6, *si.shape)).to(DEV).to(si.dtype), (0.001 * rnd(7, *qi.shape)).to(DEV).to(qi.dtype)
    y_ref = ops().pwconv(x, w, A, B, 2, 1, True)[0]
    g_ref, = torch.autograd.grad((yi, si, qi), (xi,), (gy, gs, gq), retain_graph=True)
    junk = torch.empty(32 << 20, device=DEV)
    bad = 0
    for i in range(200):
        junk.fill_(float(i))
        y = ops().pwconv(x, w, A, B, 2, 1, True)[0]
        g, = torch.autograd.grad((yi, si, qi), (xi,), (gy, gs, gq), retain_graph=True)
        bad += int(not torch.equal(y, y_ref)) + int(not torch.equal(g, g_ref))
    assert bad == 0, '%d of 400 results differ' % bad


# data gradient WITH the act' epilogue on pwk_kernel (one-slice mode: contraction over the conv's 48 .. 112 output channels, 128 < rows
# <= 256): the tile's forward input crosses the wave's LDS scratch into the lane = channel layout -- the crossing that was NOT run-to-run
# deterministic when it was tried inside pws_kernel (that variant never entered the tree).  VERDICT r3 #5: guard the shipped
# sibling with a stress test: 200 launches of each instantiated contraction depth (3, 4, 5, 6, 7 k-blocks), every result bit-identical.
PWK_ACT_DGRAD = [
    # N, Cin (rows of gx), Cout (contraction), T, H, W, act
    (2, 216, 96, 2, 14, 14, 2),       # X3D layer-3 conv3: Swish prologue, 6 k-blocks
    (1, 160, 48, 3, 6, 6, 2),         # 3 k-blocks, 5 row tiles
    (1, 250, 80, 3, 10, 10, 1),       # 5 k-blocks, ragged row tile, ReLU
    (1, 256, 112, 2, 4, 4, 2),        # 7 k-blocks, 8 full row tiles, ONE position tile
    (2, 200, 64, 4, 8, 8, 2),         # 4 k-blocks, several tiles per workgroup
]


@pytest.mark.parametrize('N,Ci,Co,T,H,W,act', PWK_ACT_DGRAD)
def test_pwk_act_dgrad_stress_bit_repeatable(split, N, Ci, Co, T, H, W, act):
    split(6)
    x = rnd(1, N, Ci, T, H, W).to(DEV).requires_grad_(True)
    w = rnd(2, Co, Ci, 1, 1, 1, scale=(2.0 / Ci) ** 0.5).to(DEV)
    A, B = (1 + 0.2 * rnd(3, N, Ci)).to(DEV).requires_grad_(True), (0.3 * rnd(4, N, Ci)).to(DEV).requires_grad_(True)
    y, s, q = ops().pwconv(x, w, A, B, act, 1, True)
    gy, gs, gq = rnd(5, *y.shape).to(DEV), (0.01 * rnd(6, *s.shape)).to(DEV).to(s.dtype), (0.001 * rnd(7, *q.shape)).to(DEV).to(q.dtype)
    ref = torch.autograd.grad((y, s, q), (x, A, B), (gy, gs, gq), retain_graph=True)
    # against fp64 first: the thing that repeats must also be right
    zd = x.detach().double() * A.detach().double().view(N, Ci, 1, 1, 1) + B.detach().double().view(N, Ci, 1, 1, 1)
    sg = torch.sigmoid(zd)
    dact = sg * (1 + zd * (1 - sg)) if act == 2 else (zd > 0).double()
    act_z = zd * sg if act == 2 else zd.clamp(min=0)
    wd = w.double().view(Co, Ci)
    yd = torch.einsum('ncthw,kc->nkthw', act_z, wd)
    gp = gy.double() + gs.double().view(N, Co, 1, 1, 1) + 2.0 * yd * gq.double().view(N, Co, 1, 1, 1)
    dz = torch.einsum('nkthw,kc->ncthw', gp, wd) * dact
    assert relerr(ref[0].double(), dz * A.detach().double().view(N, Ci, 1, 1, 1)) <= 2e-6
    assert relerr(ref[1].double(), (dz * x.detach().double()).sum((2, 3, 4))) <= 1e-5
    bad = 0
    for _ in range(200):
        g = torch.autograd.grad((y, s, q), (x, A, B), (gy, gs, gq), retain_graph=True)
        bad += int(not torch.equal(g[0], ref[0]))
        assert torch.allclose(g[1], ref[1], rtol=1e-6, atol=1e-9) and torch.allclose(g[2], ref[2], rtol=1e-6, atol=1e-9)   # fp64 atomics: order only
    assert bad == 0, '%d of 200 launches differ' % bad


def test_pwk_is_the_kernel_that_runs():
    """pwk_kernel only runs when the 6-term split is selected: with the fp32-MFMA arithmetic requested the same call goes to
    pw_deep_kernel, and the two results agree to fp32 rounding but differ in the low bits"""
    import cfn_hip
    x, w = rnd(1, 1, 216, 2, 14, 14).to(DEV), rnd(2, 96, 216, 1, 1, 1, scale=0.1).to(DEV)
    prev = cfn_hip.query('cfn_pw_split_terms', -1)
    try:
        cfn_hip.query('cfn_pw_split_terms', 6)
        y6, _, _ = ops().pwconv(x, w, None, None, 0, 1, True)
        cfn_hip.query('cfn_pw_split_terms', 0)
        y0, _, _ = ops().pwconv(x, w, None, None, 0, 1, True)
    finally:
        cfn_hip.query('cfn_pw_split_terms', prev)
    assert relerr(y6, y0) <= 3e-6 and not torch.equal(y6, y0)


# ---- staged split-bf16 weight gradient (csrc/pwsplitw.hip pws_wgrad_staged_kernel): bit-repeat stress in a launch sequence that looks
# like a backward pass.  Round 3 shipped a build of this kernel whose layer-2 conv3 weight gradient was wrong in 1-3 % of backward passes
# (in a wave that only staged operands while the other wave of its SIMD multiplied, ONE packed FMA behind an LDS coefficient read
# lost a term in lanes 48-63); two-run repeatability checks and 2,400 isolated launches never saw it.  These cases cover what
# VERDICT r4 #2 listed as uncovered: the benchmark's own layer-2/3/4 shapes at 8 clips x 256 frames, tile counts that leave waves
# without a tile (4 and 6 tiles: the `uneven` barrier path), the coarse stream's odd volume 65 x 7 x 7 (per-element strip masks),
# and the bf16 instantiation.  Between launches other kernels run on changing data (the data gradient of the same conv, a second
# conv shape, an L2-sized elementwise pass), so the kernel under test starts cold every time as it does inside a model.
WGRAD_STRESS = [
    # N, Cin, Cout, T, H, W, act, launches
    (8, 108, 48, 256, 28, 28, 2, 200),      # layer-2 conv3 at the benchmark's size (8 tiles: one per wave)
    (8, 48, 108, 256, 28, 28, 0, 200),      # layer-2 conv1
    (8, 216, 96, 256, 14, 14, 2, 200),      # layer-3 conv3 (21 tiles)
    (8, 96, 216, 256, 14, 14, 0, 200),      # layer-3 conv1
    (8, 432, 192, 256, 7, 7, 2, 200),       # layer-4 conv3 (84 tiles in groups)
    (8, 192, 432, 256, 7, 7, 0, 200),       # layer-4 conv1
    (2, 96, 48, 16, 28, 28, 2, 300),        # 2 x 3 = 6 tiles on 8 waves: two waves without a tile
    (2, 64, 64, 16, 28, 28, 1, 300),        # 2 x 2 = 4 tiles: four waves without a tile
    (2, 48, 48, 16, 14, 14, 2, 300),        # 4 tiles, short strips
    (2, 160, 96, 8, 14, 14, 2, 300),        # 3 x 5 = 15 tiles: the second tile slot is uneven
    (2, 432, 192, 65, 7, 7, 2, 300),        # coarse stream layer 4: odd volume 65 x 7 x 7 (rows start on 4-byte boundaries)
    (2, 192, 432, 65, 7, 7, 0, 300),
    (3, 216, 96, 17, 7, 7, 1, 300),         # odd volume, T = 17 (the reference's T = 64 coarse clip)
]


def _wgrad_stress(N, Cin, Cout, T, H, W, act, launches, dtype):
    o = ops()
    x = rnd(1, N, Cin, T, H, W).to(DEV).to(dtype).requires_grad_(True)
    w = rnd(2, Cout, Cin, 1, 1, 1, scale=(2.0 / Cin) ** 0.5).to(DEV).requires_grad_(True)
    A = (1 + 0.2 * rnd(3, N, Cin)).to(DEV).requires_grad_(True)
    B = (0.3 * rnd(4, N, Cin)).to(DEV).requires_grad_(True)
    y, s, q = o.pwconv(x, w, A, B, act, 1, True)
    gy = rnd(5, *y.shape).to(DEV).to(dtype)
    gs, gq = (0.01 * rnd(6, *s.shape)).to(DEV).to(s.dtype), (0.001 * rnd(7, *q.shape)).to(DEV).to(q.dtype)
    # a second conv (another kernel variant) and an L2-sized buffer to run in between
    x2 = rnd(8, 2, 54, 4, 28, 28).to(DEV).to(dtype).requires_grad_(True)
    w2 = rnd(9, 24, 54, 1, 1, 1, scale=0.2).to(DEV).requires_grad_(True)
    y2, s2, q2 = o.pwconv(x2, w2, None, None, 0, 1, True)
    junk = torch.zeros(48 << 20, device=DEV)
    ref = torch.autograd.grad((y, s, q), (w,), (gy, gs, gq), retain_graph=True)[0].clone()
    # the thing that repeats must also be right (fp64 reference of the weight gradient on a subsample of rows)
    if dtype == torch.float32:
        xd = x.detach().double()
        z = xd * A.detach().double().view(N, Cin, 1, 1, 1) + B.detach().double().view(N, Cin, 1, 1, 1)
        a = z * torch.sigmoid(z) if act == 2 else (z.clamp(min=0) if act == 1 else z)
        rows = torch.arange(0, Cout, max(Cout // 6, 1), device=DEV)
        yd = torch.einsum('ncthw,kc->nkthw', a, w.detach().double().view(Cout, Cin)[rows])
        gp = gy.double()[:, rows] + gs.double()[:, rows].view(N, -1, 1, 1, 1) + 2.0 * yd * gq.double()[:, rows].view(N, -1, 1, 1, 1)
        gw = torch.einsum('nkthw,ncthw->kc', gp, a)
        assert relerr(ref.view(Cout, Cin)[rows].double(), gw) <= 3e-6
        del xd, z, a, yd, gp
    bad = 0
    for i in range(launches):
        junk.add_(1.0)
        torch.autograd.grad((y2, s2, q2), (x2, w2), (y2.detach(), s2.detach(), q2.detach()), retain_graph=True)
        g = torch.autograd.grad((y, s, q), (w, x) if i % 4 == 0 else (w,), (gy, gs, gq), retain_graph=True)[0]
        bad += int(not torch.equal(g, ref))
    assert bad == 0, '%d of %d launches differ' % (bad, launches)


@pytest.mark.parametrize('N,Cin,Cout,T,H,W,act,launches', WGRAD_STRESS)
def test_staged_wgrad_stress_bit_repeatable(split, N, Cin, Cout, T, H, W, act, launches):
    split(6)
    _wgrad_stress(N, Cin, Cout, T, H, W, act, launches, torch.float32)


@pytest.mark.parametrize('N,Cin,Cout,T,H,W,act,launches', [c for c in WGRAD_STRESS if (c[3] * c[4] * c[5]) % 4 == 0 and c[0] < 8]
                         + [(4, 108, 48, 64, 28, 28, 2, 200), (4, 216, 96, 64, 14, 14, 2, 200), (4, 432, 192, 64, 7, 7, 2, 200)])
def test_staged_wgrad_stress_bit_repeatable_bf16(N, Cin, Cout, T, H, W, act, launches):
    _wgrad_stress(N, Cin, Cout, T, H, W, act, launches, torch.bfloat16)
