"""The hot-path kernels as registered torch operators (cfn_hip/torchlib.py): schemas on CPU; on the GPU the operators agree with
the autograd Functions of cfn_hip.ops (same C-ABI entry points) and pass torch.library.opcheck (schema, fake tensors, autograd
registration)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr, maxdiff

DEV = 'cuda'


def rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_operators_are_registered_with_schemas():
    import cfn_hip.torchlib  # noqa: F401
    assert str(torch.ops.cfn.dwconv3d.default._schema) == ('cfn::dwconv3d(Tensor x, Tensor w, Tensor? A=None, Tensor? B=None, '
                                                           'SymInt act=0, SymInt stride=1) -> (Tensor, Tensor, Tensor)')
    assert str(torch.ops.cfn.time_sample.default._schema) == 'cfn::time_sample(Tensor x, Tensor cdf) -> Tensor'
    for name in ('dwconv3d', 'dwconv3d_backward', 'pwconv', 'pwconv_backward', 'time_sample', 'time_sample_backward'):
        assert hasattr(torch.ops.cfn, name)
    # fake (meta) implementation: shapes without touching a device
    x, w = torch.empty(2, 6, 4, 14, 14, device='meta'), torch.empty(6, 1, 3, 3, 3, device='meta')
    y, s, q = torch.ops.cfn.dwconv3d(x, w, None, None, 0, 2)
    assert tuple(y.shape) == (2, 6, 4, 7, 7) and tuple(s.shape) == (2, 6) and s.dtype == torch.float64


@pytest.mark.gpu
@pytest.mark.parametrize('stride,H', [(1, 14), (2, 28), (1, 10)])
def test_dwconv3d_operator_matches_ops_and_reference(stride, H):
    import cfn_hip.torchlib  # noqa: F401
    from cfn_hip import ops
    N, C, T = 2, 5, 6
    x, w = rnd(1, N, C, T, H, H), rnd(2, C, 1, 3, 3, 3, scale=0.3)
    A, B = 1 + 0.2 * rnd(3, N, C), 0.3 * rnd(4, N, C)
    lc = [v.clone().requires_grad_(True) for v in (x, w, A, B)]
    lo = [v.clone().to(DEV).requires_grad_(True) for v in (x, w, A, B)]
    lf = [v.clone().to(DEV).requires_grad_(True) for v in (x, w, A, B)]
    a = F.relu(lc[0] * lc[2].view(N, C, 1, 1, 1) + lc[3].view(N, C, 1, 1, 1))
    yc = F.conv3d(a, lc[1], stride=(1, stride, stride), padding=1, groups=C)
    yo, so, qo = torch.ops.cfn.dwconv3d(lo[0], lo[1], lo[2], lo[3], 1, stride)
    yf, sf, qf = ops.dwconv3d(lf[0], lf[1], lf[2], lf[3], 1, stride, True)
    assert relerr(yo, yc) <= 2e-5 and torch.equal(yo, yf) and relerr(so, sf) <= 1e-12
    r, rs = rnd(7, *yc.shape), rnd(8, N, C).double()
    ((yc * r).sum() + (yc.double().sum((2, 3, 4)) * rs).sum()).backward()
    ((yo * r.to(DEV)).sum() + (so * rs.to(DEV)).sum()).backward()
    for nm, c, g in zip('xwAB', lc, lo):
        assert relerr(g.grad, c.grad) <= 2e-4, nm


@pytest.mark.gpu
def test_pwconv_and_time_sample_operators():
    import cfn_hip.torchlib  # noqa: F401
    from cfn_hip import ops
    N, Cin, Cout, T, H = 2, 24, 54, 3, 8
    x, w = rnd(1, N, Cin, T, H, H), rnd(2, Cout, Cin, 1, 1, 1, scale=0.2)
    lc = [v.clone().requires_grad_(True) for v in (x, w)]
    lo = [v.clone().to(DEV).requires_grad_(True) for v in (x, w)]
    yc = F.conv3d(lc[0], lc[1])
    yo, so, qo = torch.ops.cfn.pwconv(lo[0], lo[1])
    assert relerr(yo, yc) <= 2e-5
    r = rnd(7, *yc.shape)
    (yc * r).sum().backward()
    (yo * r.to(DEV)).sum().backward()
    for c, g in zip(lc, lo):
        assert relerr(g.grad, c.grad) <= 2e-4
    # Grid Pool resampler: operator == autograd Function (values and both gradients)
    xs = rnd(3, 2, 4, 16, 5, 5).to(DEV)
    cdf = torch.sort(torch.rand(2, 9, generator=torch.Generator().manual_seed(4)), dim=1)[0].to(DEV)
    a = [xs.clone().requires_grad_(True), cdf.clone().requires_grad_(True)]
    b = [xs.clone().requires_grad_(True), cdf.clone().requires_grad_(True)]
    oa, ob = torch.ops.cfn.time_sample(*a), ops.time_sample(*b)
    assert torch.equal(oa, ob)
    g = torch.randn_like(oa)
    (oa * g).sum().backward(); (ob * g).sum().backward()
    assert torch.equal(a[0].grad, b[0].grad) and relerr(a[1].grad, b[1].grad) <= 1e-6


@pytest.mark.gpu
def test_opcheck():
    import cfn_hip.torchlib  # noqa: F401
    x = rnd(1, 1, 3, 4, 14, 14).to(DEV).requires_grad_(True)
    w = rnd(2, 3, 1, 3, 3, 3, scale=0.3).to(DEV).requires_grad_(True)
    A, B = (1 + 0.2 * rnd(3, 1, 3)).to(DEV), (0.3 * rnd(4, 1, 3)).to(DEV)
    torch.library.opcheck(torch.ops.cfn.dwconv3d.default, (x, w, A, B, 1, 1),
                          test_utils=('test_schema', 'test_faketensor', 'test_autograd_registration'))
    xs = rnd(3, 1, 2, 8, 3, 3).to(DEV).requires_grad_(True)
    cdf = torch.linspace(0, 1, 5).view(1, 5).to(DEV).requires_grad_(True)
    torch.library.opcheck(torch.ops.cfn.time_sample.default, (xs, cdf),
                          test_utils=('test_schema', 'test_faketensor', 'test_autograd_registration'))


def test_full_operator_set_is_registered():
    """the op set behind the module API, each with a schema and a fake implementation (CPU: meta tensors only)"""
    import cfn_hip.torchlib as tl
    assert set(tl.OPERATORS) >= {'dwconv3d', 'pwconv', 'time_sample', 'dwconv_t5', 'stem_conv', 'conv3d_dense', 'bn_fold',
                                 'bn_add_relu', 'affine_act', 'pool_hw', 'interp1d', 'grid_cdf', 'gauss_align', 'fusion_gather',
                                 'film', 'time_resize'}
    for name in tl.OPERATORS:
        assert hasattr(torch.ops.cfn, name) and hasattr(torch.ops.cfn, name + '_backward'), name
    m = lambda *s, dt=torch.float32: torch.empty(*s, device='meta', dtype=dt)
    assert torch.ops.cfn.stem_conv(m(2, 3, 4, 32, 32), m(24, 3, 1, 3, 3)).shape == (2, 24, 4, 16, 16)
    y, s, q = torch.ops.cfn.conv3d_dense(m(1, 3, 8, 16, 16), m(3, 3, 3, 3, 3), [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert y.shape == (1, 3, 4, 8, 8) and s.dtype == torch.float64
    out = torch.ops.cfn.bn_fold(m(2, 6, dt=torch.float64), m(2, 6, dt=torch.float64), m(6), m(6), m(6), m(6), m((), dt=torch.int64),
                                True, 2, 6, 1, 64.0, 1e-5, 0.1)
    assert len(out) == 12 and out[0].shape == (2, 6) and out[0].dtype == torch.float64 and out[9].shape == (6,)
    assert torch.ops.cfn.time_resize(m(1, 5, 7, 3), 21, True).shape == (1, 5, 21, 3)
    assert torch.ops.cfn.grid_cdf(m(2, 16)).shape == (2, 17)
    assert torch.ops.cfn.gauss_align(m(2, 4, dt=torch.int64), m(2, 12), None, 1.0, 4.0, 1, 5).shape == (2, 12, 5)
    z, den = torch.ops.cfn.fusion_gather(m(2, 8, 12, 49), m(2, 12, 49), None, m(2, 12, 5), m(2, 12))
    assert z.shape == (2, 8, 5, 49) and den.shape == (2, 5, 49)


def _cases():
    """(operator, args, kwargs) samples for opcheck / comparison with cfn_hip.ops"""
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    N, C = 2, 6
    coef = lambda: ((1 + 0.2 * torch.randn(N, C, generator=g)).to(DEV), (0.2 * torch.randn(N, C, generator=g)).to(DEV))
    A, B = coef()
    Ar, Br = coef()
    cdf = torch.sort(torch.rand(2, 9, generator=g), dim=1)[0].to(DEV)
    s = torch.randn(N, C, generator=g).double().to(DEV) * 10
    q = (s * s / 64 + torch.rand(N, C, generator=g).double().to(DEV) * 64)
    bufs = lambda: (torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV))
    se = (r(8, C, 1, 1, 1, sc=0.3), r(8, sc=0.1), r(C, 8, 1, 1, 1, sc=0.3), r(C, sc=0.1))
    meta = torch.tensor([[2, 8, 12, 1], [0, 8, 10, 1]], dtype=torch.int64, device=DEV)
    mask = torch.ones(2, 12, device=DEV)
    return [
        ('dwconv_t5', (r(N, C, 7, 6, 6), r(C, 1, 5, 1, 1, sc=0.4)), {}),
        ('stem_conv', (r(N, 3, 3, 16, 16), r(24, 3, 1, 3, 3, sc=0.3)), {}),
        ('conv3d_dense', (r(1, 3, 8, 16, 16), r(3, 3, 3, 3, 3, sc=0.3), [3, 3, 3], [2, 2, 2], [1, 1, 1]), {}),
        ('conv3d_dense', (r(N, C, 4, 8, 8), r(1, C, 1, 3, 3, sc=0.3), [1, 3, 3], [1, 2, 2], [0, 1, 1], A, B, 1), {}),
        ('bn_fold', (s, q, r(C), r(C), *bufs(), True, N, C, 1, 64.0, 1e-5, 0.1), {}),
        ('bn_fold', (s, q, r(C), r(C), *bufs(), True, N, C, 1, 64.0, 1e-5, 0.1, *se, 64.0), {}),
        ('bn_fold', (s, None, r(C), r(C), *bufs(), False, N, C, 1, 64.0, 1e-5, 0.1, *se, 64.0), {}),
        ('bn_add_relu', (r(N, C, 3, 8, 8), A, B, r(N, C, 3, 8, 8)), {}),
        ('bn_add_relu', (r(N, C, 3, 8, 8), A, B, r(N, C, 3, 8, 8), Ar, Br), {}),
        ('affine_act', (r(N, C, 3, 8, 8), A, B, 2), {}),
        ('pool_hw', (r(N, C, 3, 14, 14), 7, 7, A, B, 1), {}),
        ('pool_hw', (r(N, C, 3, 14, 14), 1, 1), {}),
        ('interp1d', (cdf, torch.linspace(0, 1, 9).view(1, 9).repeat(2, 1).to(DEV), torch.rand(2, 9, generator=g).to(DEV)), {}),
        ('grid_cdf', (r(2, 16), r(1, sc=0.1)), {}),
        ('gauss_align', (meta, mask, cdf[:, :5].contiguous(), 1.0, 4.0, 1, 5), {}),
        ('fusion_gather', (torch.relu(r(2, 8, 12, 49)), r(2, 12, 49), r(1, sc=0.1), torch.softmax(r(2, 12, 5), 1), mask, 1), {}),
        ('film', (r(N, C, 3, 14, 14), r(N, C, 3, 7, 7), r(N, C, 3, 7, 7), 2), {}),
        ('time_resize', (r(2, 5, 7, 3), 21, True), {}),
        ('time_resize', (r(2, 5, 7), 30, False), {}),
    ]


@pytest.mark.gpu
def test_opcheck_every_operator():
    """torch.library.opcheck: schema (no undeclared mutation / aliasing), fake implementation vs real outputs, autograd
    registration -- for every registered operator, on the sample arguments above (floating-point inputs require grad)"""
    import cfn_hip.torchlib  # noqa: F401
    for name, args, kw in _cases():
        args = tuple(a.clone().requires_grad_(True) if torch.is_tensor(a) and a.is_floating_point() and a.dim() > 0 and name != 'bn_fold'
                     else a for a in args)
        if name == 'bn_fold':      # running statistics / counters are plain buffers
            args = tuple(a.clone().requires_grad_(True) if (torch.is_tensor(a) and a.is_floating_point() and i not in (4, 5)) else a
                         for i, a in enumerate(args))
        torch.library.opcheck(getattr(torch.ops.cfn, name).default, args, kw,
                              test_utils=('test_schema', 'test_autograd_registration', 'test_faketensor'))


@pytest.mark.gpu
def test_operators_equal_the_ops_functions():
    """values and gradients of torch.ops.cfn.* == cfn_hip.ops.* (same kernels): bitwise for the tensors, to fp32 rounding for
    gradients that the plain path casts at the end of the pass"""
    import cfn_hip.torchlib  # noqa: F401
    from cfn_hip import ops
    pairs = {
        'dwconv_t5': lambda x, w: ops.dwconv_t5(x, w, True),
        'stem_conv': ops.stem_conv,
        'conv3d_dense': lambda x, w, k, s, p, A=None, B=None, act=0: ops.conv3d_dense(x, w, tuple(k), tuple(s), tuple(p), A, B, act, True),
        'bn_add_relu': ops.bn_add_relu, 'affine_act': ops.affine_act,
        'pool_hw': lambda x, OH, OW, A=None, B=None, act=0: ops.pool_hw(x, OH, OW, A, B, act),
        'interp1d': ops.interp1d, 'grid_cdf': ops.grid_cdf, 'gauss_align': ops.gauss_align,
        'fusion_gather': ops.fusion_gather, 'film': ops.film, 'time_resize': ops.time_resize,
    }
    for name, args, kw in _cases():
        if name not in pairs:
            continue
        def leaves():
            return [a.clone().requires_grad_(True) if torch.is_tensor(a) and a.is_floating_point() else a for a in args]
        la, lb = leaves(), leaves()
        oa = getattr(torch.ops.cfn, name)(*la)
        ob = pairs[name](*lb)
        oa = oa if isinstance(oa, (tuple, list)) else (oa,)
        ob = ob if isinstance(ob, (tuple, list)) else (ob,)
        loss_a = loss_b = 0.0
        for k, (ta, tb) in enumerate(zip(oa, ob)):
            if tb is None:
                continue
            assert torch.equal(ta, tb), (name, k)
            if ta.is_floating_point() and ta.requires_grad:
                wgt = torch.randn(ta.shape, generator=torch.Generator().manual_seed(k)).to(DEV).to(ta.dtype)
                loss_a = loss_a + (ta * wgt).sum()
                loss_b = loss_b + (tb * wgt).sum()
        loss_a.backward()
        loss_b.backward()
        for k, (xa, xb) in enumerate(zip(la, lb)):
            if torch.is_tensor(xa) and xa.requires_grad and xb.grad is not None:
                assert xa.grad is not None, (name, k)
                assert relerr(xa.grad, xb.grad) <= 1e-6, (name, k, relerr(xa.grad, xb.grad))


@pytest.mark.gpu
@pytest.mark.parametrize('train', [False, True])
def test_bottleneck_on_torch_ops_compiles_without_graph_breaks(train, monkeypatch):
    """torch.compile(fullgraph=True) of one Bottleneck forward on torch.ops.cfn.*: dynamo traces the block into ONE graph (the
    operators are opaque nodes with fake implementations; `aot_eager` = dynamo + functionalisation + the real kernels, no
    Triton involved); the compiled block equals the eager one, and the default ctypes path"""
    import x3d_fine
    from oracle import spec
    torch._dynamo.reset()
    m = x3d_fine.Bottleneck(24, (54, 24), 1, None, index=0, base_bn_splits=1)
    spec.fill_module_(m)
    m = m.to(DEV).train(train)
    x = F.relu(spec.rand_input(3, (2, 24, 4, 8, 8))).to(DEV)
    monkeypatch.setattr(x3d_fine, 'USE_TORCH_OPS', False)
    with torch.no_grad():
        y_default = m(x)
    monkeypatch.setattr(x3d_fine, 'USE_TORCH_OPS', True)
    rm0 = m.bn2.split_bn.running_mean.clone()
    with torch.no_grad():
        y_eager = m(x)
    cm = torch.compile(m, fullgraph=True, backend='aot_eager')
    with torch.no_grad():
        y_comp = cm(x)
    assert torch.equal(y_eager, y_comp)
    assert maxdiff(y_eager, y_default) <= 1e-6
    if train:
        assert not torch.equal(m.bn2.split_bn.running_mean, rm0)      # the running statistics did move (three times)
    # and with autograd through the compiled graph
    xg = x.clone().requires_grad_(True)
    cm(xg).square().sum().backward()
    xe = x.clone().requires_grad_(True)
    m(xe).square().sum().backward()
    assert relerr(xg.grad, xe.grad) <= 1e-5


# ---- native registration: TORCH_LIBRARY inside a shared library (csrc/torch/cfn_torch.cpp -> cfn_hip/libcfn_torch.so) -------------
NATIVE_OPS = ('dwconv3d', 'dwconv3d_backward', 'pwconv', 'pwconv_backward', 'time_sample', 'time_sample_backward')


def test_native_library_defines_the_hot_path_operators():
    """the library exists, loads next to libcfn_hip.so, and the six operators carry a native kernel under the CUDA (= HIP) dispatch key whose
    schema is the one the Python definitions declare; everything else of the operator set is still registered (from Python)"""
    import os
    from cfn_hip import torchlib
    assert os.path.exists(torchlib.NATIVE_LIB), 'run python __graft_entry__.py (build_torch_library)'
    if os.environ.get('CFN_NATIVE_OPS', '1') == '0':
        pytest.skip('native operators switched off')
    assert torchlib.NATIVE
    for name in NATIVE_OPS:
        assert torch._C._dispatch_has_kernel_for_dispatch_key('cfn::' + name, 'CUDA'), name
        assert isinstance(getattr(torchlib, name), torchlib._NativeOp), name          # not a Python custom_op
    assert str(torch.ops.cfn.pwconv_backward.default._schema) == (
        'cfn::pwconv_backward(Tensor gy, Tensor gs, Tensor gq, Tensor x, Tensor w, Tensor y, Tensor? A=None, Tensor? B=None, SymInt act=0, '
        'SymInt stride=1) -> (Tensor, Tensor, Tensor, Tensor)')
    assert hasattr(torch.ops.cfn, 'bn_add_relu') and hasattr(torch.ops.cfn, 'fusion_gather')


@pytest.mark.gpu
def test_native_operators_equal_the_ctypes_route_bit_for_bit():
    """the native operators and cfn_hip.ops (ctypes) call the same C-ABI entry points on the same stream: identical bits, forward and backward,
    fp32 and bf16 / fp16 depthwise; errors of the C ABI surface as RuntimeError"""
    from cfn_hip import ops, torchlib
    if not torchlib.NATIVE:
        pytest.skip('native operators switched off')
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x = rnd(1, 2, 6, 5, 14, 14).to(DEV).to(dt).requires_grad_(True)
        w = (0.3 * rnd(2, 6, 1, 3, 3, 3)).to(DEV).requires_grad_(True)
        A, B = (1 + 0.2 * rnd(3, 2, 6)).to(DEV).requires_grad_(True), (0.1 * rnd(4, 2, 6)).to(DEV).requires_grad_(True)
        outs = []
        for route in (lambda: torch.ops.cfn.dwconv3d(x, w, A, B, 1, 1), lambda: ops.dwconv3d(x, w, A.double(), B.double(), 1, 1, True)):
            y, s, q = route()
            g = torch.autograd.grad((y.float().square().sum() + s.sum() + 0.1 * q.sum(),), (x, w, A, B))
            outs.append([y, s, q] + list(g))
        for a, b in zip(*outs):
            assert torch.equal(a.float(), b.float()) or relerr(a.double(), b.double()) <= 1e-6      # (fp64 -> fp32 casts of the coefficient gradients differ in where they happen)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][3], outs[1][3])          # y and gx: same kernels, same bits
    x = rnd(5, 2, 48, 3, 8, 8).to(DEV).requires_grad_(True)
    w = (0.2 * rnd(6, 108, 48, 1, 1, 1)).to(DEV).requires_grad_(True)
    y1, s1, q1 = torch.ops.cfn.pwconv(x, w, None, None, 0, 1)
    y2, s2, q2 = ops.pwconv(x, w, None, None, 0, 1, True)
    assert torch.equal(y1, y2) and torch.equal(s1, s2) and torch.equal(q1, q2)
    g1 = torch.autograd.grad((y1.square().sum(),), (x, w))
    g2 = torch.autograd.grad((y2.square().sum(),), (x, w))
    assert torch.equal(g1[0], g2[0]) and relerr(g1[1], g2[1]) <= 1e-6
    with pytest.raises(RuntimeError):
        torch.ops.cfn.pwconv(x.double(), w, None, None, 0, 1)
    with pytest.raises(RuntimeError):
        torch.ops.cfn.time_sample(x.cpu(), torch.rand(2, 5))
