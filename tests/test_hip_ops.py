"""GPU parity of every HIP kernel (through the C ABI / cfn_hip.ops) against plain fp32 torch on the
CPU computing the same op, forward and backward.  Tolerances are written per test: fp32 kernels,
different summation order => 1e-5 .. 1e-4 relative; index outputs bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import maxdiff, relerr, load_golden, t

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def ops():
    from cfn_hip import ops as o
    return o


def act_ref(z, act):
    if act == 1:
        return F.relu(z)
    if act == 2:
        return z * torch.sigmoid(z)
    return z


def rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def prologue_ref(x, A, B, act):
    if A is None:
        return x
    shp = A.shape + (1,) * (x.dim() - 2)
    return act_ref(x * A.view(shp) + B.view(shp), act)


def stats_ref(y):
    d = tuple(range(2, y.dim()))
    return y.double().sum(d), (y.double() * y.double()).sum(d)


def check_conv(hip_fn, ref_fn, x, w, A, B, act, tol_f=2e-5, tol_g=2e-4, x_grad=True):
    """hip_fn(x,w,A,B) -> (y,s,q) on GPU ; ref_fn(a, w) -> y on CPU with a = prologue(x)."""
    leaves_c = [v.clone().requires_grad_(True) if v is not None else None for v in (x, w, A, B)]
    leaves_g = [v.clone().to(DEV).requires_grad_(True) if v is not None else None for v in (x, w, A, B)]
    if not x_grad:
        leaves_c[0].requires_grad_(False)
        leaves_g[0].requires_grad_(False)
    yc = ref_fn(prologue_ref(leaves_c[0], leaves_c[2], leaves_c[3], act), leaves_c[1])
    sc, qc = stats_ref(yc)
    yg, sg, qg = hip_fn(*leaves_g)
    assert relerr(yg, yc) <= tol_f, ('fwd', relerr(yg, yc))
    assert relerr(sg, sc) <= 1e-5 and relerr(qg, qc) <= 1e-5, ('stats', relerr(sg, sc), relerr(qg, qc))
    r, rs, rq = rnd(7, *yc.shape), rnd(8, *sc.shape).double(), rnd(9, *qc.shape).double() * 0.1
    ((yc * r).sum() + (sc * rs).sum() + (qc * rq).sum()).backward()
    ((yg * r.to(DEV)).sum() + (sg * rs.to(DEV)).sum() + (qg * rq.to(DEV)).sum()).backward()
    names = ('x', 'w', 'A', 'B')
    for nm, c, g in zip(names, leaves_c, leaves_g):
        if c is None or not c.requires_grad:
            continue
        assert g.grad is not None, nm
        assert relerr(g.grad, c.grad) <= tol_g, ('grad ' + nm, relerr(g.grad, c.grad))


DW_CASES = [
    # N, C, T, H, W, stride, act, with_prologue
    (2, 6, 5, 7, 7, 1, 1, True),
    (1, 75, 6, 7, 7, 1, 1, True),        # more channels than one workgroup takes at 7x7
    (1, 20, 9, 14, 14, 1, 1, True),
    (1, 8, 5, 14, 14, 2, 1, True),
    (2, 5, 6, 28, 28, 1, 1, True),
    (1, 4, 4, 28, 28, 2, 0, False),
    (1, 3, 37, 56, 56, 1, 1, True),      # several t-chunks
    (1, 2, 3, 112, 112, 2, 1, True),
    (2, 5, 7, 56, 56, 2, 1, True),       # fused stride-2 backward, several bands, ReLU prologue
    (1, 3, 20, 28, 28, 2, 1, True),
    (1, 3, 4, 40, 40, 1, 1, True),       # X3D-S sizes (HS=4 path)
    (1, 2, 3, 80, 80, 2, 1, True),
    (1, 3, 3, 10, 10, 1, 2, True),       # HS=1 path
    (2, 3, 2, 5, 5, 2, 1, True),
    (1, 2, 1, 7, 7, 1, 1, True),         # T = 1
]


@pytest.mark.parametrize('N,C,T,H,W,stride,act,pro', DW_CASES)
def test_dwconv3d(N, C, T, H, W, stride, act, pro):
    x, w = rnd(1, N, C, T, H, W), rnd(2, C, 1, 3, 3, 3, scale=0.3)
    A = (1 + 0.2 * rnd(3, N, C)) if pro else None
    B = 0.3 * rnd(4, N, C) if pro else None
    check_conv(lambda x_, w_, A_, B_: ops().dwconv3d(x_, w_, A_, B_, act, stride, True),
               lambda a, w_: F.conv3d(a, w_, stride=(1, stride, stride), padding=1, groups=C), x, w, A, B, act)


WAVE_CASES = [
    # N, C, T, H, stride, act, prologue  -- the planes of the column-pair wave kernels (dwcp / dwcpb / dwcpb2)
    (2, 3, 1, 56, 1, 1, True), (1, 2, 2, 56, 1, 0, False), (1, 3, 61, 28, 1, 1, True), (2, 2, 7, 14, 1, 0, True),
    (1, 5, 4, 7, 1, 1, True), (1, 3, 3, 7, 1, 0, False),
    (1, 2, 1, 112, 2, 1, True), (2, 3, 5, 56, 2, 0, False), (1, 4, 58, 28, 2, 1, True),
    # flat kernels (dwflat.hip): 8-frame items with a ragged last item, exactly one item, the 4-frame variant (T < 12)
    (1, 2, 13, 14, 1, 1, True), (2, 2, 8, 56, 1, 1, True), (1, 2, 11, 28, 1, 0, True), (1, 2, 12, 112, 2, 1, True), (2, 2, 19, 28, 2, 1, True),
    # flat 7x7 backward (dwflatb.hip): 8-frame items, ragged last item; 14 -> 7 flat forward
    (1, 4, 21, 7, 1, 1, True), (2, 3, 16, 7, 1, 0, True), (1, 3, 13, 14, 2, 1, True),
]


@pytest.mark.parametrize('N,C,T,H,stride,act,pro', WAVE_CASES)
@pytest.mark.parametrize('stats', [True, False])
def test_dwconv3d_wave_kernels(N, C, T, H, stride, act, pro, stats):
    """T = 1 / 2 / one frame more than a t-chunk, batch > 1, no prologue, and -- stats=False -- the backward WITHOUT the
    statistics gradients (gs, gq null: the kernels' HASY = false instantiation, which the model path never takes)"""
    x, w = rnd(1, N, C, T, H, H), rnd(2, C, 1, 3, 3, 3, scale=0.3)
    A = (1 + 0.2 * rnd(3, N, C)) if pro else None
    B = 0.3 * rnd(4, N, C) if pro else None
    ref = lambda a, w_: F.conv3d(a, w_, stride=(1, stride, stride), padding=1, groups=C)
    if stats:
        check_conv(lambda x_, w_, A_, B_: ops().dwconv3d(x_, w_, A_, B_, act, stride, True), ref, x, w, A, B, act)
        return
    leaves_c = [v.clone().requires_grad_(True) if v is not None else None for v in (x, w, A, B)]
    leaves_g = [v.clone().to(DEV).requires_grad_(True) if v is not None else None for v in (x, w, A, B)]
    yc = ref(prologue_ref(leaves_c[0], leaves_c[2], leaves_c[3], act), leaves_c[1])
    yg, sg, qg = ops().dwconv3d(*leaves_g, act, stride, False)
    assert sg is None and qg is None and relerr(yg, yc) <= 2e-5
    r = rnd(7, *yc.shape)
    (yc * r).sum().backward()
    (yg * r.to(DEV)).sum().backward()
    for nm, c, g in zip(('x', 'w', 'A', 'B'), leaves_c, leaves_g):
        if c is not None:
            assert relerr(g.grad, c.grad) <= 2e-4, (nm, relerr(g.grad, c.grad))


PW_CASES = [
    # N, Cin, Cout, T, H, W, stride, act, pro
    (2, 24, 54, 3, 8, 8, 1, 0, False),
    (1, 54, 24, 4, 9, 7, 1, 2, True),
    (2, 48, 108, 2, 6, 6, 1, 1, True),
    (1, 108, 48, 3, 5, 5, 1, 2, True),
    (1, 96, 216, 2, 7, 7, 1, 1, True),
    (1, 216, 96, 2, 7, 7, 1, 2, True),
    (1, 192, 432, 3, 4, 4, 1, 0, False),
    (1, 432, 192, 2, 3, 3, 1, 2, True),
    (2, 24, 48, 3, 8, 8, 2, 1, True),     # shortcut conv, spatial stride 2
    (1, 24, 24, 2, 7, 7, 2, 0, False),
    (2, 48, 96, 3, 16, 16, 2, 2, True),    # stride 2, gathered-operand weight gradient with Swish prologue
    (1, 432, 2048, 5, 1, 1, 1, 0, False),  # fc1
    (2, 2048, 157, 3, 1, 1, 1, 1, True),   # fc2 as pointwise
    (1, 3, 5, 40, 13, 11, 1, 1, True),     # odd everything, many position tiles
    (2, 192, 432, 5, 7, 7, 1, 1, True),    # odd volume (245 positions: rows start on 4-byte boundaries), the coarse stream's layer 4
    (2, 432, 192, 13, 7, 7, 1, 2, True),
]


@pytest.mark.parametrize('N,Cin,Cout,T,H,W,stride,act,pro', PW_CASES)
def test_pwconv(N, Cin, Cout, T, H, W, stride, act, pro):
    x, w = rnd(1, N, Cin, T, H, W), rnd(2, Cout, Cin, 1, 1, 1, scale=(2.0 / Cin) ** 0.5)
    A = (1 + 0.2 * rnd(3, N, Cin)) if pro else None
    B = 0.3 * rnd(4, N, Cin) if pro else None
    check_conv(lambda x_, w_, A_, B_: ops().pwconv(x_, w_, A_, B_, act, stride, True),
               lambda a, w_: F.conv3d(a, w_, stride=(1, stride, stride)), x, w, A, B, act, tol_f=3e-5, tol_g=3e-4)


@pytest.mark.parametrize('N,C,T,H,W', [(2, 5, 7, 6, 6), (1, 24, 40, 12, 12), (1, 3, 3, 5, 7), (1, 2, 1, 8, 8), (2, 3, 70, 8, 8), (1, 2, 2, 16, 20),
                                       (1, 2, 33, 28, 28), (3, 2, 8, 20, 20)])
def test_dwconv_t5(N, C, T, H, W):
    x, w = rnd(1, N, C, T, H, W), rnd(2, C, 1, 5, 1, 1, scale=0.4)
    check_conv(lambda x_, w_, A_, B_: ops().dwconv_t5(x_, w_, True),
               lambda a, w_: F.conv3d(a, w_, padding=(2, 0, 0), groups=C), x, w, None, None, 0)


@pytest.mark.parametrize('N,T,H,W', [(2, 3, 16, 16), (1, 5, 30, 22), (1, 2, 17, 15), (1, 2, 64, 48), (2, 1, 24, 40), (1, 3, 10, 12),
                                     (1, 1, 224, 224), (2, 3, 224, 224)])
def test_stem_conv(N, T, H, W):
    x, w = rnd(1, N, 3, T, H, W), rnd(2, 24, 3, 1, 3, 3, scale=0.3)
    wc, wg = w.clone().requires_grad_(True), w.clone().to(DEV).requires_grad_(True)
    yc = F.conv3d(x, wc, stride=(1, 2, 2), padding=(0, 1, 1))
    yg = ops().stem_conv(x.to(DEV), wg)
    assert relerr(yg, yc) <= 2e-5
    r = rnd(5, *yc.shape)
    (yc * r).sum().backward()
    (yg * r.to(DEV)).sum().backward()
    assert relerr(wg.grad, wc.grad) <= 1e-4


@pytest.mark.parametrize('affine_res', [False, True])
@pytest.mark.parametrize('shape', [(2, 6, 3, 8, 8), (1, 5, 2, 7, 7), (2, 3, 5, 48, 48)])     # the last: two 8192-element chunks per channel, the second ragged
def test_bn_add_relu(shape, affine_res):
    N, C = shape[:2]
    vals = dict(y=rnd(1, *shape), A=1 + 0.2 * rnd(2, N, C), B=0.2 * rnd(3, N, C), res=rnd(4, *shape))
    if affine_res:
        vals.update(Ar=1 + 0.3 * rnd(5, N, C), Br=0.1 * rnd(6, N, C))
    c = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    g = {k: v.clone().to(DEV).requires_grad_(True) for k, v in vals.items()}
    shp = (N, C, 1, 1, 1)
    res_c = c['res'] * c['Ar'].view(shp) + c['Br'].view(shp) if affine_res else c['res']
    oc = F.relu(c['y'] * c['A'].view(shp) + c['B'].view(shp) + res_c)
    og = ops().bn_add_relu(g['y'], g['A'], g['B'], g['res'], g.get('Ar'), g.get('Br'))
    assert relerr(og, oc) <= 1e-6
    r = rnd(9, *shape)
    (oc * r).sum().backward()
    (og * r.to(DEV)).sum().backward()
    for k in vals:
        assert relerr(g[k].grad, c[k].grad) <= 2e-5, k


@pytest.mark.parametrize('N,Cin,Cout,T,H,W', [(2, 24, 24, 3, 8, 8), (1, 24, 24, 5, 14, 14), (1, 20, 32, 2, 6, 6), (3, 32, 12, 1, 4, 4),
                                              (1, 24, 24, 37, 28, 28)])
@pytest.mark.parametrize('two', [True, False])
def test_pwconv_few_channel_data_gradient(N, Cin, Cout, T, H, W, two):
    """cfn_pwconv_bwd_data_acc without prologue / epilogue at <= 32 channels on both sides (the compact shortcut gradient of layer 1,
    ops.ShortcutToken): gx = W^T (gsc gy + gs + 2 gq y) against fp64, bit-repeatable.  (A streaming VALU kernel for this shape was
    measured in round 4: 0.476 ms against pw_gemm_kernel's ~0.38 ms in the step, same-box A/B +0.1 ms per step -- not kept.)"""
    import cfn_hip
    f64 = lambda seed, *shape, scale=1.0: (rnd(seed, *shape) * scale).double().to(DEV)
    gy, y = rnd(1, N, Cout, T, H, W).to(DEV), rnd(2, N, Cout, T, H, W).to(DEV)
    w = (0.3 * rnd(4, Cout, Cin)).to(DEV)
    gs, gq, gsc = f64(5, N, Cout, scale=0.05), (f64(6, N, Cout, scale=0.01) if two else None), 1.0 + f64(7, N, Cout, scale=0.3)
    outs = []
    for _ in range(2):
        gx = torch.empty(N, Cin, T, H, W, device=DEV)
        cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y if two else None, gs, gq, w, None, None, None, 0, gx, None, None, N, Cin, Cout, T, H, W, 1,
                     None, 1, gsc)
        outs.append(gx)
    assert torch.equal(outs[0], outs[1])
    gp = gy.double() * gsc.view(N, Cout, 1, 1, 1) + gs.view(N, Cout, 1, 1, 1)
    if two:
        gp = gp + 2.0 * gq.view(N, Cout, 1, 1, 1) * y.double()
    ref = torch.einsum('nkthw,km->nmthw', gp, w.double())
    assert relerr(outs[0].double(), ref) <= 2e-6


@pytest.mark.parametrize('N,Cin,Cout,T,H,W,s', [(2, 96, 432, 3, 14, 14, 2),     # layer-4 block 0 conv1: streamed-weight split-bf16 kernel + lattice add
                                                (1, 48, 216, 2, 28, 28, 2),     # layer-3 block 0 conv1: pwk_kernel + lattice add
                                                (1, 96, 432, 2, 7, 9, 2),       # odd plane width: 4 x 5 lattice
                                                (2, 48, 216, 1, 6, 6, 3),       # stride 3
                                                (1, 54, 108, 2, 8, 8, 2)])      # layer-2 shape: stays on the in-kernel lattice loads
@pytest.mark.parametrize('two', [True, False])
def test_pwconv_data_gradient_with_compact_shortcut_gradient(N, Cin, Cout, T, H, W, s, two):
    """cfn_pwconv_bwd_data_acc WITHOUT act' epilogue and WITH the compact gradient of a strided shortcut conv (first block of a stage):
    gx = W^T (gsc gy + gs + 2 gq y) + acc on the lattice h % s == w % s == 0, against fp64, bit-repeatable.  The deep shapes run the
    contraction on a split-bf16 kernel and add the lattice behind it (csrc/pwconv.hip pw_lattice_add_kernel)."""
    import cfn_hip
    f64 = lambda seed, *shape, scale=1.0: (rnd(seed, *shape) * scale).double().to(DEV)
    gy, y = rnd(1, N, Cout, T, H, W).to(DEV), rnd(2, N, Cout, T, H, W).to(DEV)
    w = ((2.0 / Cout) ** 0.5 * rnd(4, Cout, Cin)).to(DEV)
    gs, gq, gsc = f64(5, N, Cout, scale=0.05), (f64(6, N, Cout, scale=0.01) if two else None), 1.0 + f64(7, N, Cout, scale=0.3)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    acc = rnd(8, N, Cin, T, Ho, Wo).to(DEV)
    outs = []
    for _ in range(3):
        gx = torch.full((N, Cin, T, H, W), float('nan'), device=DEV)
        cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y if two else None, gs, gq, w, None, None, None, 0, gx, None, None, N, Cin, Cout, T, H, W, 1,
                     acc, s, gsc)
        outs.append(gx)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    gp = gy.double() * gsc.view(N, Cout, 1, 1, 1) + gs.view(N, Cout, 1, 1, 1)
    if two:
        gp = gp + 2.0 * gq.view(N, Cout, 1, 1, 1) * y.double()
    ref = torch.einsum('nkthw,km->nmthw', gp, w.double())
    ref[:, :, :, ::s, ::s] += acc.double()
    assert relerr(outs[0].double(), ref) <= 2e-6


@pytest.mark.parametrize('act', [None, 0, 1, 2])
@pytest.mark.parametrize('cfg', [(2, 24, 54, 3, 8, 8, 0), (2, 54, 24, 2, 12, 12, 0), (1, 24, 24, 5, 6, 6, 2), (2, 64, 32, 1, 10, 10, 0),
                                 (1, 20, 12, 3, 6, 6, 2), (2, 24, 54, 4, 8, 8, 2), (1, 3, 7, 2, 4, 6, 0)])
def test_pwconv_bwd_fused_matches_separate(cfg, act):
    """cfn_pwconv_bwd_fused (one pass: data + weight gradient) against cfn_pwconv_bwd_data_acc + cfn_pwconv_bwd_weight on the
    same inputs: with / without prologue (act None = no A,B), statistics gradients, the tail scale `gscale` and the compact
    shortcut gradient `acc` on its stride lattice; ragged position counts (not a multiple of the 64-position stage)."""
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

    def run(fused):
        gx = torch.empty_like(x)
        gA = gB = None
        if A is not None:
            gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        a_ = 0 if act is None else act
        if fused:
            ok = cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, gw, N, Cin, Cout, T, H, W,
                                  acc, acc_s or 1, gsc)
            assert ok, 'shape should be handled by the fused kernel'
        else:
            cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, N, Cin, Cout, T, H, W, 1, acc,
                         acc_s or 1, gsc)
            cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, a_, gw, N, Cin, Cout, T, H, W, 1, gsc)
        return gx, gA, gB, gw

    ref, got = run(False), run(True)
    for name, r, g in zip(('gx', 'gA', 'gB', 'gw'), ref, got):
        if r is not None:
            assert relerr(g, r) <= 2e-5, name


@pytest.mark.parametrize('act', [None, 0, 1, 2])
@pytest.mark.parametrize('cfg', [(2, 48, 108, 3, 8, 8, 0), (2, 108, 48, 2, 12, 12, 0), (1, 48, 108, 5, 6, 6, 2), (1, 108, 48, 3, 10, 10, 2),
                                 (2, 40, 100, 1, 7, 8, 0), (1, 100, 40, 2, 6, 6, 0), (1, 64, 128, 9, 14, 14, 0), (1, 128, 64, 4, 28, 28, 0),
                                 (2, 24, 108, 3, 12, 12, 0), (1, 32, 128, 2, 10, 10, 2), (1, 16, 70, 5, 6, 6, 0)])
def test_pwconv_bwd_fused_split_matches_separate(cfg, act, monkeypatch):
    """the layer-2 widths of cfn_pwconv_bwd_fused (csrc/pwfuseds.hip: one pass, split-bf16 matrix products) against the separate data /
    weight gradient kernels: prologue / no prologue, statistics gradients, tail scale, compact shortcut gradient, ragged strips"""
    import cfn_hip
    monkeypatch.setenv('CFN_PWF_SPLIT', '2')
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

    def run(fused):
        gx = torch.empty_like(x)
        gA = gB = None
        if A is not None:
            gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        a_ = 0 if act is None else act
        if fused:
            ok = cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, gw, N, Cin, Cout, T, H, W,
                                  acc, acc_s or 1, gsc)
            assert ok, 'shape should be handled by the split fused kernel'
        else:
            cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w, x, A, B, a_, gx, gA, gB, N, Cin, Cout, T, H, W, 1, acc,
                         acc_s or 1, gsc)
            cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, a_, gw, N, Cin, Cout, T, H, W, 1, gsc)
        return gx, gA, gB, gw

    ref, got = run(False), run(True)
    for name, r, g in zip(('gx', 'gA', 'gB', 'gw'), ref, got):
        if r is not None:
            assert relerr(g, r) <= 2e-5, (name, relerr(g, r))
    again = run(True)                      # run-to-run bits (fp64 atomics of fp32 partials: exact)
    for r, g in zip(got, again):
        if r is not None:
            assert torch.equal(r, g)


@pytest.mark.parametrize('two', [True, False])
@pytest.mark.parametrize('cfg', [(1, 96, 216, 3, 14, 14, 0), (2, 96, 216, 2, 14, 14, 0), (1, 90, 200, 1, 10, 10, 0), (2, 72, 196, 5, 6, 6, 0), (1, 96, 216, 16, 14, 14, 0),
                                 (1, 48, 216, 3, 14, 14, 0), (2, 48, 216, 2, 28, 28, 2), (1, 40, 200, 2, 12, 12, 2), (1, 96, 216, 2, 14, 14, 2)])
def test_pwconv_bwd_fused_split_layer3_matches_separate(cfg, two, monkeypatch):
    """the layer-3 variant of the one-pass backward (csrc/pwfuseds.hip, pw_bwd_fused_split3_kernel: W^T pre-split into a workspace and read out of L2, 6 weight-gradient
    + 2 data-gradient waves, 32-position stages) against the separate kernels; no prologue (conv1 of a block), with / without the batch-norm terms"""
    import cfn_hip
    monkeypatch.setenv('CFN_PWF_SPLIT', '1')
    N, Cin, Cout, T, H, W, acc_s = cfg
    f64 = lambda seed, *shape, scale=1.0: (rnd(seed, *shape) * scale).double().to(DEV)
    gy, y, x = rnd(1, N, Cout, T, H, W).to(DEV), rnd(2, N, Cout, T, H, W).to(DEV), rnd(3, N, Cin, T, H, W).to(DEV)
    w = (0.3 * rnd(4, Cout, Cin)).to(DEV)
    gs, gq, gsc = (f64(5, N, Cout, scale=0.05), f64(6, N, Cout, scale=0.01), 1.0 + f64(7, N, Cout, scale=0.3)) if two else (None, None, None)
    acc = rnd(10, N, Cin, T, (H - 1) // acc_s + 1, (W - 1) // acc_s + 1).to(DEV) if acc_s else None      # compact shortcut gradient of a stage-first block

    def run(fused):
        gx = torch.full_like(x, float('nan'))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        if fused:
            ok = cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, None, None, 0, gx, None, None, gw, N, Cin, Cout, T, H, W, acc, acc_s or 1, gsc)
            assert ok, 'shape should be handled by the layer-3 split fused kernel'
        else:
            cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w, x, None, None, 0, gx, None, None, N, Cin, Cout, T, H, W, 1, acc, acc_s or 1, gsc)
            cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, None, None, 0, gw, N, Cin, Cout, T, H, W, 1, gsc)
        return gx, gw

    ref, got, again = run(False), run(True), run(True)
    for name, r, g, g2 in zip(('gx', 'gw'), ref, got, again):
        assert relerr(g, r) <= 2e-5, (name, relerr(g, r))
        assert torch.equal(g, g2), name                # run-to-run bits


@pytest.mark.parametrize('act', [0, 1, 2])
@pytest.mark.parametrize('cfg', [(1, 216, 96, 3, 14, 14), (2, 216, 96, 2, 14, 14), (1, 200, 90, 1, 10, 10), (2, 196, 72, 5, 6, 6), (1, 216, 96, 16, 14, 14)])
def test_pwconv_bwd_fused_split_layer3_conv3_matches_separate(cfg, act, monkeypatch):
    """the conv3 side of the layer-3 one-pass backward (pw_bwd_fused_split3e_kernel: 216 -> 96 behind a prologue; act' epilogue and the statistics of 216 channels in the
    data-gradient waves) against the separate kernels: gx, gA, gB, gw; launched twice: identical bits"""
    import cfn_hip
    monkeypatch.setenv('CFN_PWF_L3E', '1')
    N, Cin, Cout, T, H, W = cfg
    f64 = lambda seed, *shape, scale=1.0: (rnd(seed, *shape) * scale).double().to(DEV)
    gy, y, x = rnd(1, N, Cout, T, H, W).to(DEV), rnd(2, N, Cout, T, H, W).to(DEV), rnd(3, N, Cin, T, H, W).to(DEV)
    w = (0.3 * rnd(4, Cout, Cin)).to(DEV)
    gs, gq, gsc = f64(5, N, Cout, scale=0.05), f64(6, N, Cout, scale=0.01), 1.0 + f64(7, N, Cout, scale=0.3)
    A, B = 1.0 + f64(8, N, Cin, scale=0.2), f64(9, N, Cin, scale=0.2)

    def run(fused):
        gx = torch.full_like(x, float('nan'))
        gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        if fused:
            ok = cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, A, B, act, gx, gA, gB, gw, N, Cin, Cout, T, H, W, None, 1, gsc)
            assert ok, 'shape should be handled by the layer-3 conv3 split fused kernel'
        else:
            cfn_hip.call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w, x, A, B, act, gx, gA, gB, N, Cin, Cout, T, H, W, 1, None, 1, gsc)
            cfn_hip.call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, act, gw, N, Cin, Cout, T, H, W, 1, gsc)
        return gx, gA, gB, gw

    ref, got, again = run(False), run(True), run(True)
    for name, r, g, g2 in zip(('gx', 'gA', 'gB', 'gw'), ref, got, again):
        assert relerr(g, r) <= 2e-5, (name, relerr(g, r))
        assert torch.equal(g, g2), name


@pytest.mark.parametrize('cfg', [(4, 48, 108, 16, 28, 28, 1), (2, 24, 108, 8, 56, 56, None), (8, 96, 216, 8, 14, 14, None), (2, 48, 216, 8, 28, 28, None), (2, 108, 48, 8, 28, 28, 2)])
def test_pwconv_bwd_fused_split_bit_repeat_stress(cfg, monkeypatch):
    """200 launches of the one-pass kernels on the same inputs: every output bit-identical to the first launch's.  (The first 8-wave build of csrc/pwfuseds.hip reproduced the
    round-3 race here -- one G' row wrong in the rows staged by lanes 48-63 in 1-7 of 100 launches, DESIGN 4.1 -- and this is the test that would catch it again; the shapes are
    the ones tools/pwfs_stress.py saw it on, plus the layer-3 instances.)"""
    import cfn_hip
    monkeypatch.setenv('CFN_PWF_SPLIT', '2')
    N, Cin, Cout, T, H, W, act = cfg
    f64 = lambda seed, *shape, scale=1.0: (rnd(seed, *shape) * scale).double().to(DEV)
    gy, y, x = rnd(1, N, Cout, T, H, W).to(DEV), rnd(2, N, Cout, T, H, W).to(DEV), rnd(3, N, Cin, T, H, W).to(DEV)
    w = (0.3 * rnd(4, Cout, Cin)).to(DEV)
    gs, gq, gsc = f64(5, N, Cout, scale=0.05), f64(6, N, Cout, scale=0.01), 1.0 + f64(7, N, Cout, scale=0.3)
    A = B = None
    if act is not None:
        A, B = 1.0 + f64(8, N, Cin, scale=0.2), f64(9, N, Cin, scale=0.2)

    def run():
        gx = torch.full_like(x, float('nan'))
        gA = gB = None
        if A is not None:
            gA, gB = (torch.zeros(N, Cin, dtype=torch.float64, device=DEV) for _ in range(2))
        gw = torch.zeros(Cout, Cin, dtype=torch.float64, device=DEV)
        assert cfn_hip.call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w, x, A, B, act or 0, gx, gA, gB, gw, N, Cin, Cout, T, H, W, None, 1, gsc)
        return [v for v in (gx, gA, gB, gw) if v is not None]

    ref = run()
    for i in range(200):
        for a_, b_ in zip(run(), ref):
            assert torch.equal(a_, b_), 'launch %d differs' % i


@pytest.mark.parametrize('shortcut', ['identity', 'conv_s2', 'conv_s1'])
@pytest.mark.parametrize('cfg', [(2, 54, 24, 4, 8, 8), (1, 108, 48, 3, 6, 6), (2, 216, 96, 2, 14, 14), (1, 20, 12, 3, 5, 7)])
def test_linked_tail(cfg, shortcut):
    """conv3 -> (stats) -> tail with a TailLink: the tail's backward hands ONE unscaled gradient tensor to conv3 and to the
    shortcut conv, whose kernels apply the per-(n,c) factors (`gscale`); the forward emits the ReLU bit mask (or, for
    volumes that are not a multiple of 4, the backward falls back to `out`).  Reference: plain torch autograd on the CPU."""
    N, Cm, Co, T, H, W = cfg
    s = 1 if shortcut != 'conv_s2' else 2
    Hi, Wi = H * s, W * s
    Cx = Co if shortcut == 'identity' else 10
    vals = dict(x2=rnd(1, N, Cm, T, H, W), w3=0.2 * rnd(2, Co, Cm, 1, 1, 1), A2=1 + 0.2 * rnd(3, N, Cm), B2=0.1 * rnd(4, N, Cm),
                A3=1 + 0.3 * rnd(5, N, Co), B3=0.2 * rnd(6, N, Co), xr=rnd(7, N, Cx, T, Hi, Wi))
    if shortcut != 'identity':
        vals.update(wd=0.3 * rnd(8, Co, Cx, 1, 1, 1), Ad=1 + 0.3 * rnd(9, N, Co), Bd=0.1 * rnd(10, N, Co))
    vals['A3'][0, :3] = 0.0     # zero BN scale (gamma == 0): the weight-gradient kernels divide by a floored scale
    vals['A3'][-1, -1] = -0.5
    c = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    g = {k: v.clone().to(DEV).requires_grad_(True) for k, v in vals.items()}

    def ref(d):
        shp = (N, -1, 1, 1, 1)
        a2 = prologue_ref(d['x2'], d['A2'], d['B2'], 2)
        y3 = F.conv3d(a2, d['w3'])
        s3, q3 = stats_ref(y3)
        z = y3 * d['A3'].view(shp) + d['B3'].view(shp)
        if shortcut == 'identity':
            r = d['xr']
            extra = 0.0
        else:
            yd = F.conv3d(d['xr'], d['wd'], stride=(1, s, s))
            sd, qd = stats_ref(yd)
            r = yd * d['Ad'].view(shp) + d['Bd'].view(shp)
            extra = (sd * 0.02).sum() + (qd * 0.003).sum()
        return F.relu(z + r), (s3 * 0.03).sum() + (q3 * 0.004).sum() + extra

    def hip(d):
        o = ops()
        link = o.TailLink()
        y3, s3, q3 = o.pwconv(d['x2'], d['w3'], d['A2'], d['B2'], 2, 1, True, tail=link, tail_role='y')
        if shortcut == 'identity':
            out = o.bn_add_relu(y3, d['A3'], d['B3'], d['xr'], link=link)
            extra = 0.0
        else:
            yd, sd, qd = o.pwconv(d['xr'], d['wd'], None, None, 0, s, True, tail=link, tail_role='res')
            out = o.bn_add_relu(y3, d['A3'], d['B3'], yd, d['Ad'], d['Bd'], link=link)
            extra = (sd * 0.02).sum() + (qd * 0.003).sum()
        return out, (s3 * 0.03).sum() + (q3 * 0.004).sum() + extra

    oc, ec = ref(c)
    og, eg = hip(g)
    assert relerr(og, oc) <= 1e-5
    r = rnd(11, *oc.shape)
    ((oc * r).sum() + ec).backward()
    ((og * r.to(DEV)).sum() + eg.float()).backward()
    for k in vals:
        assert relerr(g[k].grad, c[k].grad) <= 5e-5, k


@pytest.mark.parametrize('act', [0, 1, 2])
def test_affine_act_and_stats(act):
    shape = (2, 5, 3, 6, 6)
    x, A, B = rnd(1, *shape), 1 + 0.2 * rnd(2, 2, 5), 0.2 * rnd(3, 2, 5)
    c = [v.clone().requires_grad_(True) for v in (x, A, B)]
    g = [v.clone().to(DEV).requires_grad_(True) for v in (x, A, B)]
    oc = prologue_ref(c[0], c[1], c[2], act)
    og = ops().affine_act(g[0], g[1], g[2], act)
    assert relerr(og, oc) <= 1e-6
    sg, qg = ops().channel_stats(g[0])
    sc, qc = stats_ref(c[0])
    assert relerr(sg, sc) <= 1e-6 and relerr(qg, qc) <= 1e-6
    r = rnd(9, *shape)
    ((oc * r).sum() + (sc * 0.3).sum() + (qc * 0.1).sum()).backward()
    ((og * r.to(DEV)).sum() + (sg * 0.3).sum() + (qg * 0.1).sum()).backward()
    for a, b in zip(g, c):
        assert relerr(a.grad, b.grad) <= 2e-5


@pytest.mark.parametrize('H,W,OH,OW,pro', [(7, 7, 1, 1, True), (14, 14, 7, 7, False), (16, 16, 7, 7, False),
                                           (8, 8, 7, 7, True), (7, 7, 7, 7, True), (56, 56, 7, 7, False),
                                           (7, 7, 1, 1, False), (8, 8, 1, 1, True), (5, 5, 1, 1, True), (14, 14, 1, 1, True)])
@pytest.mark.parametrize('T', [3, 41])       # 41 frames: several frame groups per wave of the global-mean kernel, ragged tail
def test_pool_hw(H, W, OH, OW, pro, T):
    shape = (2, 4, T, H, W)
    x = rnd(1, *shape)
    A = 1 + 0.2 * rnd(2, 2, 4) if pro else None
    B = 0.2 * rnd(3, 2, 4) if pro else None
    c = [v.clone().requires_grad_(True) if v is not None else None for v in (x, A, B)]
    g = [v.clone().to(DEV).requires_grad_(True) if v is not None else None for v in (x, A, B)]
    oc = F.adaptive_avg_pool3d(prologue_ref(c[0], c[1], c[2], 1), (None, OH, OW))
    og = ops().pool_hw(g[0], OH, OW, g[1], g[2], 1 if pro else 0)
    assert relerr(og, oc) <= 2e-6
    r = rnd(9, *oc.shape)
    (oc * r).sum().backward()
    (og * r.to(DEV)).sum().backward()
    for a, b in zip(g, c):
        if a is not None:
            assert relerr(a.grad, b.grad) <= 2e-5


def test_film():
    x, m, cc = rnd(1, 2, 3, 4, 14, 14), rnd(2, 2, 3, 4, 7, 7), rnd(3, 2, 3, 4, 7, 7)
    c = [v.clone().requires_grad_(True) for v in (x, m, cc)]
    g = [v.clone().to(DEV).requires_grad_(True) for v in (x, m, cc)]
    up = lambda v: v.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    oc = c[0] * up(c[1]) + up(c[2])
    og = ops().film(g[0], g[1], g[2], 2)
    assert relerr(og, oc) <= 1e-6
    r = rnd(9, *oc.shape)
    (oc * r).sum().backward()
    (og * r.to(DEV)).sum().backward()
    for a, b in zip(g, c):
        assert relerr(a.grad, b.grad) <= 2e-5


# ---- Grid Pool / Unpool / Interp1d --------------------------------------------------------------------
def test_grid_time_index_bit_exact_on_reference_vectors():
    """indices the reference's grid_sample used (golden, captured from the reference run)"""
    for name, T in (('gridsample_ulp', 16), ('gridpool_d24_eval', 64), ('gridpool_d24_train', 64),
                    ('gridpool_d4_eval', 16), ('gridpool_d4_train', 16)):
        z = load_golden(name)
        i0, _ = ops().grid_time_index(t(z['cdf']).to(DEV), T)
        assert torch.equal(i0.cpu(), t(z['i0'])), name


def test_grid_time_index_bit_exact_random_sweep():
    from oracle import x3d_ref as R
    g = torch.Generator().manual_seed(5)
    for T in (16, 64, 256, 1000):
        p = torch.rand(64, 65, generator=g) + 0.05
        cdf = torch.cumsum((p / p.sum(1, keepdim=True)).double(), 1).float()
        cdf = torch.cat([torch.zeros(64, 1), cdf], 1)
        i0c, w1c = R.grid_sample_time_index(cdf, T)
        i0g, w1g = ops().grid_time_index(cdf.to(DEV), T)
        assert torch.equal(i0g.cpu(), i0c)
        assert torch.equal(w1g.cpu(), w1c)


def test_time_sample_matches_grid_sample_and_grads():
    from oracle import x3d_ref as R
    z = load_golden('gridsample_ulp')
    x, cdf = t(z['x']), t(z['cdf'])
    xc, cc = x.clone().requires_grad_(True), cdf.clone().requires_grad_(True)
    xg, cg = x.clone().to(DEV).requires_grad_(True), cdf.clone().to(DEV).requires_grad_(True)
    yc = R.grid_pool_resample(xc, cc)
    yg = ops().time_sample(xg, cg)
    # 2-tap temporal lerp vs 8-tap trilinear whose h/w coordinates are integers up to 1 ulp: 2e-5 abs
    assert maxdiff(yg, z['y']) <= 2e-5 and maxdiff(yg, yc) <= 2e-5
    r = rnd(3, *yc.shape)
    (yc * r).sum().backward()
    (yg * r.to(DEV)).sum().backward()
    assert relerr(xg.grad, xc.grad) <= 2e-5
    assert relerr(cg.grad, cc.grad) <= 1e-4


def test_time_sample_large_plane_vec4():
    from oracle import x3d_ref as R
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, 32, 12, 12, generator=g)
    p = torch.rand(1, 8, generator=g) + 0.1
    cdf = torch.cat([torch.zeros(1, 1), torch.cumsum(p / p.sum(), 1)], 1)
    yc = R.grid_pool_resample(x, cdf)
    yg = ops().time_sample(x.to(DEV), cdf.to(DEV))
    assert maxdiff(yg, yc) <= 2e-5


@pytest.mark.parametrize('k', [5, 17, 65])
def test_interp1d_bit_exact(k):
    z = load_golden('interp1d_k%d' % k)
    y, ind = ops().interp1d(t(z['x']).to(DEV), t(z['mid']).to(DEV), t(z['mid']).to(DEV))
    assert torch.equal(ind.cpu(), t(z['ind']))
    assert torch.equal(y.cpu(), t(z['ynew']))
    y2, ind2 = ops().interp1d(t(z['x']).to(DEV), t(z['y2']).to(DEV), t(z['q2']).to(DEV))
    assert torch.equal(ind2.cpu(), t(z['ind2']))
    assert maxdiff(y2, z['ynew2']) == 0.0


def test_interp1d_grads():
    from oracle import x3d_ref as R
    z = load_golden('interp1d_k17')
    vals = [t(z['x']), t(z['y2']), t(z['q2'])]
    c = [v.clone().requires_grad_(True) for v in vals]
    g = [v.clone().to(DEV).requires_grad_(True) for v in vals]
    yc, _ = R.interp1d(*c)
    yg, _ = ops().interp1d(*g)
    r = rnd(2, *yc.shape)
    (yc * r).sum().backward()
    (yg * r.to(DEV)).sum().backward()
    for a, b in zip(g, c):
        assert relerr(a.grad, b.grad) <= 1e-5


@pytest.mark.parametrize('shape,L', [((2, 7, 17), 64), ((1, 3, 17, 6, 6), 68), ((1, 2, 5), 16)])
def test_time_resize(shape, L):
    x = rnd(1, *shape)
    xc, xg = x.clone().requires_grad_(True), x.clone().to(DEV).requires_grad_(True)
    if len(shape) == 3:
        yc = F.interpolate(xc, L, mode='linear', align_corners=True)
    else:
        yc = F.interpolate(xc, (L,) + shape[3:], mode='trilinear', align_corners=True)
    yg = ops().time_resize(xg, L)
    assert maxdiff(yg, yc) <= 1e-6
    r = rnd(2, *yc.shape)
    (yc * r).sum().backward()
    (yg * r.to(DEV)).sum().backward()
    assert relerr(xg.grad, xc.grad) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('shape,L', [((2, 7, 16), 160), ((3, 157, 64), 640), ((1, 2, 5), 3), ((1, 3, 9), 9)])
def test_time_resize_half_pixel(shape, L):
    """align_corners=False: the loss upsampling of train_coarse_fineFEAT.py:226 (ATen upsample_linear1d)"""
    x = rnd(1, *shape)
    xc, xg = x.clone().requires_grad_(True), x.clone().to(DEV).requires_grad_(True)
    yc = F.interpolate(xc, L, mode='linear')
    yg = ops().time_resize(xg, L, False)
    assert maxdiff(yg, yc) <= 1e-6
    r = rnd(2, *yc.shape)
    (yc * r).sum().backward()
    (yg * r.to(DEV)).sum().backward()
    assert relerr(xg.grad, xc.grad) <= 1e-5


def test_ops_refuse_cpu_tensors():
    """the product path has no CPU fallback"""
    with pytest.raises(RuntimeError):
        ops().dwconv3d(torch.zeros(1, 2, 2, 7, 7), torch.zeros(2, 1, 3, 3, 3))


# ---- dense implicit-GEMM conv (Grid Pool saliency convs) and the fusion gather ---------------------------
DENSE_CASES = [
    # N, Cin, Cout, T, H, W, kernel, stride, padding, act, pro
    (2, 4, 4, 8, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1), 0, False),
    (1, 24, 24, 6, 14, 14, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, True),
    (1, 24, 1, 4, 14, 14, (1, 3, 3), (1, 2, 2), (0, 1, 1), 1, True),
    (2, 3, 24, 3, 16, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), 0, False),
    (1, 5, 7, 5, 9, 7, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, True),
    (1, 24, 24, 6, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, True),   # Wo % 4 == 0: direct im2col weight gradient
    (2, 6, 40, 5, 16, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), 1, True),
    # the Grid Pool saliency shapes at their real planes (csrc/salconv.hip): conv1 56 -> 28 without prologue, conv2 28 -> 14 with
    # the ReLU prologue; odd / even / single frame counts, several t-chunks, a ragged last band (28-wide: 14 output rows in bands of 8)
    (2, 24, 24, 9, 56, 56, (3, 3, 3), (2, 2, 2), (1, 1, 1), 0, False),
    (1, 24, 24, 12, 28, 28, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, True),
    (1, 24, 24, 1, 56, 56, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, True),
    (1, 24, 20, 34, 28, 28, (3, 3, 3), (2, 2, 2), (1, 1, 1), 0, False),
    (1, 24, 24, 5, 20, 56, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, True),
]


@pytest.mark.parametrize('N,Ci,Co,T,H,W,k,s,p,act,pro', DENSE_CASES)
def test_conv3d_dense(N, Ci, Co, T, H, W, k, s, p, act, pro):
    x, w = rnd(1, N, Ci, T, H, W), rnd(2, Co, Ci, *k, scale=(2.0 / (Ci * k[0] * k[1] * k[2])) ** 0.5)
    A = (1 + 0.2 * rnd(3, N, Ci)) if pro else None
    B = 0.3 * rnd(4, N, Ci) if pro else None
    check_conv(lambda x_, w_, A_, B_: ops().conv3d_dense(x_, w_, k, s, p, A_, B_, act, True),
               lambda a, w_: F.conv3d(a, w_, stride=s, padding=p), x, w, A, B, act, tol_f=3e-5, tol_g=3e-4)


@pytest.mark.parametrize('B,crops,C,Tf,K,P', [(2, 1, 8, 12, 5, 49), (1, 1, 24, 40, 17, 49), (1, 1, 5, 7, 3, 1),
                                                  (2, 3, 6, 10, 5, 49)])
def test_fusion_gather(B, crops, C, Tf, K, P):
    """gather with the attention sigmoid, the mask multiply and the multi-crop repeat folded in, vs the plain torch
    expression of x3d_coarse.py:209-223 (forward 1e-5, every gradient 1e-4 relative)"""
    x, at_raw, bias = rnd(1, B, C, Tf, P).abs(), rnd(2, B, Tf, P), torch.tensor([0.3])
    GX, mask = rnd(3, B * crops, Tf, K).abs(), torch.ones(B, Tf)
    mask[:, -2:] = 0          # masked fine steps
    c = [v.clone().requires_grad_(True) for v in (x, at_raw, bias, GX)]
    g = [v.clone().to(DEV).requires_grad_(True) for v in (x, at_raw, bias, GX)]
    rep = lambda v: v.unsqueeze(1).repeat((1, crops) + (1,) * (v.dim() - 1)).view((B * crops,) + tuple(v.shape[1:]))
    at = rep(torch.sigmoid(c[1] + c[2]))
    wgt = at.unsqueeze(2) * (c[3] * rep(mask).unsqueeze(2)).unsqueeze(3)                       # B2 Tf K P
    zc = torch.einsum('bctp,btkp->bckp', rep(c[0]), wgt) / (wgt.sum(1) + 1e-6).unsqueeze(1)
    zg = ops().fusion_gather(g[0], g[1], g[2], g[3], mask.to(DEV), crops)
    assert relerr(zg, zc) <= 1e-5
    r = rnd(5, *zc.shape)
    (zc * r).sum().backward()
    (zg * r.to(DEV)).sum().backward()
    for a, b in zip(g, c):
        assert relerr(a.grad, b.grad) <= 1e-4, (a.shape, relerr(a.grad, b.grad))


@pytest.mark.parametrize('B,crops,Tf,K,T,grid', [(3, 1, 40, 17, 64, True), (2, 2, 20, 9, 32, True), (2, 1, 24, 16, None, False)])
def test_gauss_align(B, crops, Tf, K, T, grid):
    """Gaussian alignment kernel vs the reference's tensor expression (oracle.gaussian): forward 1e-6, CDF gradient 1e-4"""
    from oracle import x3d_ref as R
    g = torch.Generator().manual_seed(17)
    mask = torch.ones(B, Tf)
    mask[-1, Tf - 5:] = 0
    meta = torch.stack([torch.tensor([3 * i, 64, Tf, 1 + i]) for i in range(B)]).to(torch.int64)
    if grid:
        cdf = torch.sort(torch.rand(B * crops, K, generator=g), 1)[0]
        cc, cg = cdf.clone().requires_grad_(True), cdf.clone().to(DEV).requires_grad_(True)
        ref = R.gaussian(meta, mask, cc, T)
        out = ops().gauss_align(meta.to(DEV), mask.to(DEV), cg, T, 1, crops, K)
    else:
        ref = R.gaussian(meta, mask, torch.zeros(B, 1, K), None)
        out = ops().gauss_align(meta.to(DEV), mask.to(DEV), None, None, 1, crops, K)
    assert out.shape == ref.shape and maxdiff(out, ref) <= 1e-6
    if grid:
        r = rnd(5, *ref.shape)
        (ref * r).sum().backward()
        (out * r.to(DEV)).sum().backward()
        assert relerr(cg.grad, cc.grad) <= 1e-4


@pytest.mark.parametrize('B,Kin', [(2, 4), (3, 16), (1, 64), (5, 33)])
def test_grid_cdf_kernel(B, Kin):
    """saliency logits -> CDF knots in one kernel vs the reference expression on the CPU (x3d_coarse.py:384-392):
    values within 2 ulp of 1.0, first knot exactly 0, monotone; gradient 1e-4"""
    g0, bias = rnd(3, B, Kin, scale=2.0), torch.tensor([0.25])
    gc, bc = g0.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    gg, bg = g0.clone().to(DEV).requires_grad_(True), bias.clone().to(DEV).requires_grad_(True)
    p = 1. - torch.sigmoid((gc + bc) * 5e-1)
    p = p / (torch.sum(p, dim=1, keepdim=True) + 1e-16)
    ref = torch.cat([torch.zeros(B, 1), torch.cumsum(p, dim=1)], dim=1)
    out = ops().grid_cdf(gg, bg)
    assert maxdiff(out, ref) <= 2.5e-7
    assert float(out[:, 0].abs().max()) == 0.0 and bool((out[:, 1:] >= out[:, :-1]).all())
    r = rnd(4, B, Kin + 1)
    (ref * r).sum().backward()
    (out * r.to(DEV)).sum().backward()
    assert relerr(gg.grad, gc.grad) <= 1e-4 and relerr(bg.grad, bc.grad) <= 1e-4


def test_grid_index_mismatch_rate_vs_cpu_reference():
    """End-to-end index statement: 10k random saliency rows (T=256, K=65) through the CDF kernel + the index
    kernel against the reference's CPU expression + ATen index arithmetic.

    What is bit exact is the contract `identical CDF in -> identical indices out`.  End to end, the CDF itself is only equal
    to ~1e-7 (ATen's vectorised CPU sigmoid / sum are host-ISA specific, the GPU convs differ in the last bits), which moves
    floor(i_t) only where i_t sits within an ulp of an integer.  Interior knots: essentially never.  The LAST knot: always --
    cdf[-1] = fl(sum p) is 1-2^-24, 1 or 1+2^-23 according to the rounding error of the row sum S (SURVEY 7 measured 39 % != 1
    on the CPU itself), so i0[-1] is T-2 (weight ~1 on frame T-1) or T-1 (weight 0 on the out-of-range frame T): the same
    resampled frame, two spellings.  Measured here: interior 0 of 640k, last knot ~41 %."""
    from oracle import x3d_ref as R
    rows, Kin, T = 10000, 64, 256
    g0 = rnd(21, rows, Kin, scale=1.5)
    ref_cdf = R.grid_cdf(g0)
    i_ref, _ = R.grid_sample_time_index(ref_cdf, T)
    cdf = ops().grid_cdf(g0.to(DEV))
    i0, _ = ops().grid_time_index(cdf, T)
    mism = (i0.cpu() != i_ref)
    rate_inner = float(mism[:, :-1].float().mean())
    rate_last = float(mism[:, -1].float().mean())
    print('grid index mismatch vs CPU reference: %.2e of interior indices, %.2e of last-knot indices' % (rate_inner, rate_last))
    assert maxdiff(cdf, ref_cdf) <= 1e-6
    assert rate_inner <= 1e-4
    assert int((i0.cpu() - i_ref).abs().max()) <= 1          # a flipped index is the neighbouring frame ...
    x = rnd(22, 64, 1, T, 3)                                 # ... and resamples to the same value (continuity of the lerp)
    ya = ops().time_sample(x.to(DEV), cdf[:64].contiguous())
    yb = ops().time_sample(x.to(DEV), ref_cdf[:64].to(DEV))
    assert maxdiff(ya, yb) <= 2e-3                         # |d cdf| (1e-6) x 255 frames x |x| (~4)
    # same CDF in -> same indices out, bit for bit
    i_same, _ = ops().grid_time_index(ref_cdf.to(DEV), T)
    assert torch.equal(i_same.cpu(), i_ref)


def test_pwconv_prologue_without_activation():
    """bias-only prologue (A=1, B=b, act none) as used by the fusion MLPs: gradients w.r.t. A/B must flow"""
    N, Cin, Cout = 2, 24, 16
    x, w = rnd(1, N, Cin, 3, 7, 7), rnd(2, Cout, Cin, 1, 1, 1, scale=0.3)
    A, B = 1 + 0.2 * rnd(3, N, Cin), 0.3 * rnd(4, N, Cin)
    check_conv(lambda x_, w_, A_, B_: ops().pwconv(x_, w_, A_, B_, 0, 1, True),
               lambda a, w_: F.conv3d(a, w_), x, w, A, B, 0, tol_f=3e-5, tol_g=3e-4)


@pytest.mark.parametrize('case', ['1d_all', 'many_x_row_y', 'stacked_queries', 'row_query', 'edge_queries'])
def test_interp1d_module_shape_rules(case):
    """Interp1d()(x, y, xnew, out) call convention and broadcasting of interp1d.py:20-71 (1-D inputs, a single row
    against several, stacked query rows, the `out` buffer), values and indices bit-exact against the oracle; queries
    left / right of the knot range extrapolate from the first / last segment like the reference.  (One x row against
    several y rows with several query rows raises inside torch.searchsorted in the reference itself: not a case.)"""
    from interp1d import Interp1d
    from oracle import x3d_ref as R
    g = torch.Generator().manual_seed(11)
    N, P, Bn = 17, 23, 4
    xs = torch.sort(torch.rand(Bn, N, generator=g), 1)[0]
    ys = torch.randn(Bn, N, generator=g)
    qs = torch.rand(Bn, P, generator=g)
    if case == '1d_all':
        x, y, q = xs[0], ys[0], qs[0]
    elif case == 'many_x_row_y':
        x, y, q = xs, ys[:1], qs
    elif case == 'stacked_queries':
        x, y, q = xs[0], ys[0], qs
    elif case == 'row_query':
        x, y, q = xs, ys, qs[:1]
    else:
        x, y = xs, ys
        q = torch.cat([torch.full((Bn, 2), -0.5), xs[:, :3], xs[:, -2:], torch.full((Bn, 2), 1.5)], 1)   # outside + on the knots
    yc, indc = R.interp1d(x, y, q)
    out = torch.zeros_like(yc).to(DEV)
    yg, indg = Interp1d().forward(x.to(DEV), y.to(DEV), q.to(DEV), out=out, return_index=True)
    assert yg.shape == yc.shape
    assert torch.equal(indg.cpu().view(indc.shape), indc), case
    assert torch.equal(yg.cpu(), yc), case
    assert torch.equal(out.cpu().view(yc.shape), yc), case


@pytest.mark.parametrize('case', ['dw_s1', 'dw_s2', 'dw_small', 'pw_fused', 'pw_deep'])
def test_bitwise_reproducible(case):
    """Same inputs, three runs: outputs, statistics and gradients must be bit-identical.  Inside a workgroup every fp32
    reduction has a fixed order (per-wave slots, no LDS float atomics); across workgroups fp32 partials are accumulated in
    fp64, which is exact -- hence order independent -- as long as the partials of one sum lie within ~2^19 of each other."""
    o = ops()
    if case.startswith('dw'):
        C, T, H, s = {'dw_s1': (6, 9, 28, 1), 'dw_s2': (5, 8, 28, 2), 'dw_small': (40, 12, 7, 1)}[case]
        x, w = rnd(1, 2, C, T, H, H).to(DEV), (0.2 * rnd(2, C, 1, 3, 3, 3)).to(DEV)
        A, B = (1 + 0.2 * rnd(3, 2, C)).to(DEV), (0.1 * rnd(4, 2, C)).to(DEV)
        fn = lambda x_, w_, A_, B_: o.dwconv3d(x_, w_, A_, B_, 1, s, True)
    else:
        Ci, Co, T, H = (24, 54, 8, 16) if case == 'pw_fused' else (96, 216, 4, 14)
        x, w = rnd(1, 2, Ci, T, H, H).to(DEV), (0.2 * rnd(2, Co, Ci, 1, 1, 1)).to(DEV)
        A, B = (1 + 0.2 * rnd(3, 2, Ci)).to(DEV), (0.1 * rnd(4, 2, Ci)).to(DEV)
        fn = lambda x_, w_, A_, B_: o.pwconv(x_, w_, A_, B_, 2, 1, True)
    runs = []
    for _ in range(3):
        leaves = [v.clone().requires_grad_(True) for v in (x, w, A, B)]
        y, sm, sq = fn(*leaves)
        gy = rnd(9, *y.shape).to(DEV)
        ((y * gy).sum() + (sm * 0.01).sum().float() + (sq * 0.001).sum().float()).backward()
        runs.append([y.detach(), sm.detach(), sq.detach()] + [v.grad for v in leaves])
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('op', ['affine_act', 'channel_stats', 'bn_add_relu', 'pool_hw', 'dwconv_t5'])
def test_more_than_65535_channel_rows(op, dtype):
    """batch x channels beyond grid.y's 65535 (batch 32 of X3D-M's 2048-wide head; big validation batches): the (n, c) index
    is factorised over grid.y x grid.z.  Checked against the same op run on the two halves of the batch (each <= 65535 rows,
    the path the CPU-parity tests above cover): forward and gradients identical."""
    o = ops()
    N, C, T, H, W = 70, 1000, 2, 4, 4                    # N*C = 70000 = 35000 x 2
    dt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    if dtype == 'bf16' and op in ('affine_act', 'channel_stats'):
        pytest.skip('fp32-only op')
    x = rnd(1, N, C, T, H, W).to(DEV).to(dt)
    A, B = (1 + 0.2 * rnd(2, N, C)).to(DEV), (0.2 * rnd(3, N, C)).to(DEV)
    res = rnd(4, N, C, T, H, W).to(DEV).to(dt)
    w5 = rnd(5, C, 1, 5, 1, 1).to(DEV)

    def run(sl):
        xs = x[sl].clone().requires_grad_(True)
        As, Bs = A[sl].clone().requires_grad_(True), B[sl].clone().requires_grad_(True)
        if op == 'affine_act':
            out = (o.affine_act(xs, As, Bs, 2),)
        elif op == 'channel_stats':
            out = o.channel_stats(xs)
        elif op == 'bn_add_relu':
            out = (o.bn_add_relu(xs, As, Bs, res[sl]),)
        elif op == 'pool_hw':
            out = (o.pool_hw(xs, 1, 1, As, Bs, 1),)
        else:
            xin = xs.float() if dtype == 'bf16' else xs
            out = o.dwconv_t5(xin, w5, stats=True, out_dtype=dt if dtype == 'bf16' else None)
        sum((v.double() * (0.5 + i)).sum() for i, v in enumerate(out) if v is not None).backward()
        grads = [xs.grad] + ([As.grad, Bs.grad] if As.grad is not None else [])
        return [v.detach() for v in out if v is not None], grads

    full_o, full_g = run(slice(0, N))
    lo_o, lo_g = run(slice(0, N // 2))
    hi_o, hi_g = run(slice(N // 2, N))
    for f, a, b in zip(full_o + full_g, lo_o + lo_g, hi_o + hi_g):
        assert torch.equal(f, torch.cat([a, b], 0))


def test_shape_mismatches_raise_instead_of_reading_out_of_bounds():
    """the C ABI takes pointers and sizes: tensors whose RELATIVE shapes do not fit are refused by the ops layer (a 96 x 96 coarse clip against
    7 x 7 fine features used to fault the device inside the FiLM kernel)"""
    o = ops()
    x = rnd(1, 2, 8, 3, 24, 24).to(DEV)
    m7, c7 = rnd(2, 2, 8, 3, 7, 7).to(DEV), rnd(3, 2, 8, 3, 7, 7).to(DEV)
    with pytest.raises(RuntimeError, match='do not tile'):
        o.film(x, m7, c7, 24 // 7)
    m8 = rnd(4, 2, 8, 3, 8, 8).to(DEV)
    assert o.film(x, m8, m8, 3).shape == x.shape
    w = rnd(5, 16, 8, 1, 1, 1).to(DEV)
    with pytest.raises(RuntimeError, match='coefficients'):
        o.pwconv(x, w, torch.ones(2, 7, device=DEV), torch.zeros(2, 7, device=DEV), 0, 1, True)
    wd = rnd(6, 8, 1, 3, 3, 3).to(DEV)
    with pytest.raises(RuntimeError, match='coefficients'):
        o.dwconv3d(x, wd, torch.ones(1, 8, device=DEV), torch.zeros(1, 8, device=DEV), 0, 1, True)
    y = rnd(7, 2, 8, 3, 12, 12).to(DEV)
    with pytest.raises(RuntimeError, match='residual'):
        o.bn_add_relu(y, torch.ones(2, 8, device=DEV), torch.zeros(2, 8, device=DEV), x)


def test_double_backward_is_refused_loudly():
    """the backward passes are kernels, not differentiable graphs: asking for a second derivative raises instead of returning gradients that
    silently ignore it"""
    o = ops()
    x = rnd(1, 1, 8, 2, 8, 8).to(DEV).requires_grad_(True)
    w = rnd(2, 8, 1, 3, 3, 3).to(DEV).requires_grad_(True)
    y, _, _ = o.dwconv3d(x, w, None, None, 0, 1, True)
    gx, = torch.autograd.grad(y.sum(), x, create_graph=True)
    with pytest.raises(RuntimeError):                       # the first derivative carries no graph at all ...
        gx.sum().backward()
    gy = torch.ones_like(y).requires_grad_(True)
    gx, = torch.autograd.grad(y, x, gy, create_graph=True)
    with pytest.raises(RuntimeError, match='once_differentiable|differentiate twice'):   # ... and says so when a graph is forced through it
        gx.sum().backward()
