"""The hot-path kernels as registered torch operators (cfn_hip/torchlib.py): schemas on CPU; on the GPU the operators agree with
the autograd Functions of cfn_hip.ops (same C-ABI entry points) and pass torch.library.opcheck (schema, fake tensors, autograd
registration)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

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
