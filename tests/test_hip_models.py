"""GPU parity of the product modules (x3d_fine on the HIP path) against the golden vectors captured
from the reference and against the CPU oracle on the same seeded inputs."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, golden_sd, t, maxdiff, relerr

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def thin(v, limit=20000):
    return v if v.numel() <= limit else v.flatten()[::37]


def _load(module, sd):
    module.load_state_dict({k: v.clone() for k, v in sd.items()})
    return module.to(DEV)


@pytest.mark.parametrize('S', [1, 2])
def test_subbn_module(S):
    import x3d_fine
    from oracle import spec
    z = load_golden('subbn_s%d' % S)
    m = _load(x3d_fine.SubBatchNorm3d(num_splits=S, num_features=6, affine=True), golden_sd(z))
    x1 = (spec.rand_input(81, (4, 6, 3, 5, 5)) * 1.7 + 0.3).to(DEV)
    x2 = (spec.rand_input(82, (4, 6, 3, 5, 5)) * 0.6 - 0.2).to(DEV)
    m.train(True)
    assert maxdiff(m(x1), z['y1']) <= 5e-6
    assert maxdiff(m(x2), z['y2']) <= 5e-6
    assert maxdiff(m.split_bn.running_mean, z['split_rm']) <= 2e-6
    assert maxdiff(m.split_bn.running_var, z['split_rv']) <= 2e-6
    m.aggregate_stats()
    assert maxdiff(m.bn.running_mean, z['rm']) <= 2e-6 and maxdiff(m.bn.running_var, z['rv']) <= 2e-6
    m.train(False)
    assert maxdiff(m(x1), z['y3']) <= 5e-6


# l3_* / l4_*: layer-3 / 4 widths at their real planes (216- / 432-channel depthwise + SE, split-bf16 and fp32-MFMA pointwise kernels,
# 14x14 / 7x7 / 28->14 / 14->7 depthwise kernels): train-mode forward + backward against the REFERENCE's vectors
@pytest.mark.parametrize('tag,index,stride,cin,planes,shape', [
    ('even_s1', 0, 1, 24, (54, 24), (2, 24, 4, 8, 8)), ('odd_s1', 1, 1, 24, (54, 24), (2, 24, 4, 8, 8)),
    ('even_s2', 0, 2, 24, (54, 48), (2, 24, 4, 8, 8)), ('odd_s2', 1, 2, 48, (108, 48), (2, 48, 4, 8, 8)),
    ('l3_even_s2', 0, 2, 48, (216, 96), (2, 48, 2, 28, 28)), ('l3_odd_s1', 1, 1, 96, (216, 96), (2, 96, 2, 14, 14)),
    ('l4_even_s2', 0, 2, 96, (432, 192), (2, 96, 2, 14, 14)), ('l4_odd_s1', 1, 1, 192, (432, 192), (2, 192, 2, 7, 7))])
@pytest.mark.parametrize('torch_ops', [False, True])
def test_bottleneck_vs_reference(tag, index, stride, cin, planes, shape, torch_ops, monkeypatch):
    """torch_ops=True: the same block through the registered dispatcher operators torch.ops.cfn.* (x3d_fine.USE_TORCH_OPS)"""
    import x3d_fine
    from oracle import spec
    monkeypatch.setattr(x3d_fine, 'USE_TORCH_OPS', torch_ops)
    z = load_golden('bottleneck_' + tag)
    ds = None
    if stride != 1 or cin != planes[1]:
        ds = torch.nn.Sequential(x3d_fine.conv1x1x1(cin, planes[1], stride),
                                 x3d_fine.SubBatchNorm3d(num_splits=1, num_features=planes[1], affine=True))
    m = _load(x3d_fine.Bottleneck(cin, planes, stride, ds, index=index, base_bn_splits=1), golden_sd(z))
    m.train(True)
    x = F.relu(spec.rand_input(91, shape)).to(DEV).requires_grad_(True)
    y = m(x)
    assert maxdiff(y, z['y']) <= 2e-5
    (y * spec.rand_input(92, tuple(y.shape)).to(DEV)).sum().backward()
    assert relerr(x.grad, z['gx']) <= 2e-4
    named = dict(m.named_parameters())
    for k in z:
        if k.startswith('g_'):
            name = k[2:].replace('_weight', '.weight').replace('_bias', '.bias').replace('downsample_', 'downsample.')
            assert relerr(named[name].grad, z[k]) <= 5e-4, (k, relerr(named[name].grad, z[k]))
    assert maxdiff(m.bn2.split_bn.running_mean, z['bn2_rm']) <= 2e-6
    assert maxdiff(m.bn2.split_bn.running_var, z['bn2_rv']) <= 2e-6


def test_fine_cfg1_logits_vs_reference():
    """BASELINE.json configs[0]; north_star tolerance: logits within 1e-3 fp32"""
    import x3d_fine
    from oracle import spec
    z = load_golden('fine_cfg1')
    m = x3d_fine.generate_model('S', n_classes=400, task='loc', base_bn_splits=1)
    m.replace_logits(157)
    _load(m, golden_sd(z)).eval()
    with torch.no_grad():
        y = m([spec.rand_input(0, (1, 3, 13, 160, 160)).to(DEV), None])
    assert y.shape == (1, 157, 13)
    assert maxdiff(y, z['logits']) <= 1e-3
    assert maxdiff(y, z['logits']) <= 1e-4   # what fp32 kernels actually deliver


def test_fine_tower_vs_reference():
    import x3d_fine
    from oracle import spec
    z = load_golden('fine_tower')
    m = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, global_tower=True)
    spec.fill_module_(m)
    m.to(DEV).eval()
    with torch.no_grad():
        f, _ = m([spec.rand_input(1, (1, 3, 6, 64, 64)).to(DEV), None])
    for k in ('layer1', 'layer2', 'layer3', 'layer4', 'conv5'):
        assert maxdiff(f[k], z[k]) <= 5e-5, k


def test_fine_train_fwd_bwd_vs_reference():
    import x3d_fine
    from oracle import spec
    z = load_golden('fine_train')
    m = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(m)
    m.to(DEV).train(True)
    y = m([spec.rand_input(2, (2, 3, 8, 64, 64)).to(DEV), None])
    assert maxdiff(y, z['logits']) <= 1e-3
    (y * spec.rand_input(3, tuple(y.shape)).to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    gn = json.loads(str(z['grad_norms']))
    # Whole-net gradients of this tiny train-mode case are ill-conditioned in fp32: scaling the input clip by
    # (1 + 1e-6) moves the CPU oracle's own gradients by up to 22 % element-wise and 1.7 % in norm (measured,
    # ReLU / SE kinks behind batch-statistics BN).  So: norms within 5 %, direction (cosine) within 2 %;
    # the tight backward checks are the module-level ones (test_bottleneck_vs_reference, tests/test_hip_ops.py).
    bad = []
    for k, ref in gn.items():
        mine = float(named[k].grad.double().norm())
        if abs(mine - ref) > 5e-2 * max(ref, 1e-1):
            bad.append((k, mine, ref))
    assert not bad, bad[:10]
    for k in z:
        if k.startswith('g_'):
            name = [n for n in gn if ('g_' + n.replace('.', '_')) == k][0]
            a, b = thin(named[name].grad).detach().cpu().double().flatten(), t(z[k]).double().flatten()
            cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            assert cos >= 0.98, (k, cos)
    assert maxdiff(m.layer2[0].bn2.split_bn.running_mean, z['bn_rm']) <= 1e-5


def test_fine_eval_matches_oracle_at_224():
    """same seeded input through the HIP path and the CPU oracle, X3D-M 1x3x8x224x224 eval"""
    import x3d_fine
    from oracle import spec, x3d_ref
    m = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1)
    spec.fill_module_(m)
    m.to(DEV).eval()
    x = spec.rand_input(5, (1, 3, 8, 224, 224))
    with torch.no_grad():
        y = m([x.to(DEV), None])
        yo = x3d_ref.x3d_fine_forward(spec.procedural_fill(spec.fine_keys('M', 157, 1)), x, 'M', training=False)
    assert maxdiff(y, yo) <= 1e-3


def test_fine_eval_mode_backward_vs_oracle():
    """eval-mode (running statistics) forward + backward of X3D-M against the CPU oracle: well conditioned, so the gradients
    are compared tightly (norm-relative 2e-3), including the depthwise weights of squeeze-excite blocks, whose gradient
    has a term through the SE average pool even when the BN statistics are constants"""
    import x3d_fine
    from oracle import spec, x3d_ref
    m = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(m)
    m.to(DEV).eval()
    x = spec.rand_input(21, (1, 3, 8, 112, 112))
    y = m([x.to(DEV), None])
    r = spec.rand_input(22, tuple(y.shape))
    (y * r.to(DEV)).sum().backward()
    sd = spec.procedural_fill(spec.fine_keys('M', 157, 1))
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    yo = x3d_ref.x3d_fine_forward(sd, x, 'M', training=False)
    (yo * r).sum().backward()
    assert maxdiff(y, yo) <= 1e-3
    named = dict(m.named_parameters())
    worst = ('', 0.0)
    for k in ('conv1_s.weight', 'conv1_t.weight', 'layer1.0.conv2.weight', 'layer1.0.fc1.weight', 'layer1.1.conv2.weight',
              'layer2.0.downsample.0.weight', 'layer2.2.bn2.weight', 'layer3.4.conv3.weight', 'layer4.6.conv2.weight',
              'layer4.6.fc2.bias', 'layer4.5.conv1.weight', 'conv5.weight', 'bn5.bias', 'fc1.weight', 'fc2.weight'):
        a, b = named[k].grad.cpu().double().flatten(), sd[k].grad.double().flatten()
        e = float((a - b).norm() / b.norm())
        if e > worst[1]:
            worst = (k, e)
    print('eval-mode backward vs oracle: worst norm-rel %.2e (%s)' % (worst[1], worst[0]))
    assert worst[1] <= 2e-3, worst


# ---- coarse stream -----------------------------------------------------------------------------------
@pytest.fixture(params=[False, True], ids=['ops', 'torch_ops'])
def op_route(request, monkeypatch):
    """the module under test on cfn_hip.ops (default) and on the registered dispatcher operators torch.ops.cfn.* (CFN_USE_TORCH_OPS)"""
    import x3d_fine
    monkeypatch.setattr(x3d_fine, 'USE_TORCH_OPS', request.param)
    return request.param


@pytest.mark.parametrize('tag,depth', [('d4', 4), ('d24', 24)])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_gridpool_layer_vs_reference(tag, depth, mode, op_route):
    import x3d_coarse
    from cfn_hip import ops
    from oracle import spec
    z = load_golden('gridpool_%s_%s' % (tag, mode))
    m = _load(x3d_coarse.GridPoolLayer(4, depth), golden_sd(z))
    m.train(mode == 'train')
    x = spec.rand_input(int(z['seed']), tuple(int(v) for v in z['shape'])).to(DEV)
    with torch.no_grad():
        y, cdf = m(x)
    assert maxdiff(cdf, z['cdf']) <= 2e-6
    # index contract: identical CDF in => identical frame indices out (bit exact)
    i0, _ = ops.grid_time_index(t(z['cdf']).to(DEV), x.shape[2])
    assert torch.equal(i0.cpu(), t(z['i0']))
    # ... and the indices the layer derives from ITS OWN CDF (saliency convs + CDF kernel on the GPU) are the reference's
    # (interior knots exactly; the last knot's cdf[-1] in {1-2^-24, 1, 1+2^-23} spells the same frame as T-2 (w=1) or T-1
    # (w=0), decided by one ulp of the row sum -- tests/test_hip_ops.py::test_grid_index_mismatch_rate_vs_cpu_reference)
    i_own, _ = ops.grid_time_index(cdf, x.shape[2])
    assert torch.equal(i_own.cpu()[:, :-1], t(z['i0'])[:, :-1])
    assert int((i_own.cpu()[:, -1] - t(z['i0'])[:, -1]).abs().max()) <= 1
    assert maxdiff(y, z['y']) <= 1e-4
    if mode == 'train':
        assert maxdiff(m.bn1.split_bn.running_mean, z['rm1']) <= 1e-5
        assert maxdiff(m.bn2.split_bn.running_var, z['rv2']) <= 1e-5


def test_gridpool_conv_bias_gets_an_exact_zero_gradient_and_weight_decay():
    """x3d_coarse.py:362-366: conv1 / conv2 of Grid Pool feed train-mode batch norms, so their biases cancel (the reference computes a gradient
    of rounding noise for them and SGD's weight decay still shrinks them).  Round 6 folds the bias-free statistics directly and keeps the
    bias in the graph through `_ZeroGradFor`: the gradient must be exactly zero, the optimizer must still see the parameter (decay applied),
    and the output must not depend on the bias while the running mean does."""
    import x3d_coarse
    from oracle import spec
    torch.manual_seed(0)
    m = x3d_coarse.GridPoolLayer(4, 24)
    spec.fill_module_(m)
    m = m.to(DEV).train(True)
    x = spec.rand_input(77, (2, 24, 16, 28, 28)).to(DEV)
    b0 = m.conv1.bias.detach().clone()
    rm0 = m.bn1.split_bn.running_mean.detach().clone()
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=0.1)
    y, cdf = m(x)
    (y * spec.rand_input(78, tuple(y.shape)).to(DEV)).sum().backward()
    for conv in (m.conv1, m.conv2):
        assert conv.bias.grad is not None and not bool(conv.bias.grad.any())
        assert float(conv.weight.grad.abs().max()) > 0
    rm1 = m.bn1.split_bn.running_mean.detach().clone()
    opt.step()
    assert maxdiff(m.conv1.bias, b0 * (1 - 0.1 * 0.1)) <= 1e-7
    # the same input with a shifted bias: same output / CDF, running mean moved by momentum x shift
    m2 = x3d_coarse.GridPoolLayer(4, 24)
    spec.fill_module_(m2)
    m2 = m2.to(DEV).train(True)
    with torch.no_grad():
        m2.conv1.bias.add_(0.5)
    y2, cdf2 = m2(x)
    assert maxdiff(y2, y) <= 1e-5 and maxdiff(cdf2, cdf) <= 1e-6
    assert maxdiff(m2.bn1.split_bn.running_mean - rm1, torch.full_like(rm1, 0.5 * m.bn1.momentum)) <= 1e-6
    assert float((rm1 - rm0).abs().max()) > 0


def test_gridunpool_vs_reference(op_route):
    import x3d_coarse
    from cfn_hip import ops
    z = load_golden('gridunpool')
    cdf = t(z['cdf']).to(DEV)
    yl = x3d_coarse.GridUnpool([t(z['xl']).to(DEV), cdf, True])
    assert maxdiff(yl, z['yl']) <= 2e-6
    assert maxdiff(ops.time_resize(yl, (yl.shape[2] - 1) * 4), z['yl_up']) <= 2e-6
    yf = x3d_coarse.GridUnpool([t(z['xf']).to(DEV), cdf, False])
    assert maxdiff(yf, z['yf']) <= 2e-5


@pytest.mark.parametrize('name', ['gaussian', 'gaussian_multicrop'])
def test_gaussian_vs_reference(name, op_route):
    """Gaussian module = one HIP kernel; 'gaussian_multicrop': b2 = 2b, crop j starts at start + step*j (x3d_coarse.py:264-266)"""
    import x3d_coarse
    z = load_golden(name)
    g = x3d_coarse.Gaussian(ratio=1)([t(z['meta']).to(DEV), t(z['mask']).to(DEV), t(z['cdf']).to(DEV), int(z['T'])])
    assert tuple(g.shape) == z['GX'].shape and maxdiff(g, z['GX']) <= 1e-6


@pytest.mark.parametrize('mix', [True, False])
def test_rewight_multicrop_vs_reference(mix, op_route):
    """b2 = 2b (validation-time multi-crop, x3d_coarse.py:209-211): fine features / mask of a video are shared by its crops"""
    import x3d_coarse
    z = load_golden('rewight_multicrop_%s' % ('mix' if mix else 'nomix'))
    m = _load(x3d_coarse.RewightLayer(channels=6, g_channels=6, depth=8, height=14), golden_sd(z)).eval()
    b2, K = z['GX'].shape[0], z['GX'].shape[2]
    lx = torch.zeros(b2, 6, K, 14, 14, device=DEV)
    with torch.no_grad():
        b_, s_ = m([t(z['xf']).to(DEV), lx, t(z['mask']).to(DEV), None, 0, t(z['GX']).to(DEV), mix])
    assert b_.shape[0] == 2 * z['xf'].shape[0]
    assert maxdiff(b_, z['bias']) <= 1e-5 and maxdiff(s_, z['scale']) <= 1e-5


@pytest.mark.parametrize('name,hgt,mix,pool', [('rewight_h7_mix', 7, True, False), ('rewight_h7_nomix', 7, False, False),
                                               ('rewight_h14_mix', 14, True, False),
                                               ('rewight_h14_nomix', 14, False, False),
                                               ('rewight_pool', 7, False, True)])
def test_rewight_vs_reference(name, hgt, mix, pool, op_route):
    import x3d_coarse
    z = load_golden(name)
    ch = z['bias'].shape[1]
    m = _load(x3d_coarse.RewightLayer(channels=ch, g_channels=ch, depth=8, height=hgt, pool=pool), golden_sd(z)).eval()
    K = z['GX'].shape[2]
    lx = torch.zeros(z['xf'].shape[0], ch, K, 1 if pool else hgt, 1 if pool else hgt, device=DEV)
    with torch.no_grad():
        b_, s_ = m([t(z['xf']).to(DEV), lx, t(z['mask']).to(DEV), None, 0, t(z['GX']).to(DEV), mix])
    assert maxdiff(b_, z['bias']) <= 1e-5
    assert maxdiff(s_, z['scale']) <= 1e-5


@pytest.mark.parametrize('li,h', [(0, 14), (3, 7)])
def test_mixing_vs_reference(li, h, op_route):
    import x3d_coarse
    from oracle import spec
    z = load_golden('mixing_l%d' % li)
    chans = (24, 48, 96, 192)
    m = _load(x3d_coarse.MixingLayer(depth=chans[li], learned=True, index=li), golden_sd(z)).eval()
    B, K = 1, int(z['K'])
    bias, scale = [], []
    for j, (c, hh) in enumerate(zip(chans, (56, 28, 14, 7))):
        up = lambda v: F.adaptive_max_pool2d(v.view(B, c * K, 7, 7), (hh, hh)).view(B, c, K, hh, hh)
        bias.append(up(spec.rand_input(60 + j, (B, c, K, 7, 7))).to(DEV))
        scale.append(up(spec.rand_input(70 + j, (B, c, K, 7, 7))).to(DEV))
    with torch.no_grad():
        c_, m_ = m([torch.zeros(B, chans[li], K, h, h, device=DEV), bias, scale])
    assert maxdiff(c_, z['c']) <= 2e-5
    assert maxdiff(m_, z['m']) <= 1e-5


def _coarse_inputs(seed, B, T, Tf):
    from oracle import spec
    x = spec.rand_input(seed, (B, 3, T, 224, 224))
    depth = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}
    feat = {k: spec.rand_input(seed + 1 + i, (B, c, Tf, 7, 7), nonneg=True) for i, (k, c) in enumerate(depth.items())}
    fm = torch.ones(B, Tf)
    meta = torch.zeros(B, 4, dtype=torch.int64)
    for b in range(B):
        valid = Tf - 3 * b
        fm[b, valid:] = 0
        meta[b] = torch.tensor([b * 2, T, valid, 1])
    return x, feat, fm, meta, depth


def _coarse_model(depth, dropout=0.5):
    import x3d_coarse
    from oracle import spec
    m = x3d_coarse.generate_model('M', n_classes=400, feat_depth=depth, task='loc', dropout=dropout, base_bn_splits=1,
                                  learnedMixing=True, isMixing=True, t_pool='grid')
    m.replace_logits(157)
    spec.fill_module_(m)
    return m.to(DEV)


def test_coarse_eval_logits_vs_reference(op_route):
    """full Coarse-Fine forward (fineFEAT fusion), north_star tolerance 1e-3 on logits"""
    z = load_golden('coarse_eval')
    x, feat, fm, meta, depth = _coarse_inputs(100, 1, 16, 12)
    m = _coarse_model(depth).eval()
    with torch.no_grad():
        y = m([x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), 0, meta.to(DEV)])
        _, cdf = m.pool_1(m.layer1(m._stem(x.to(DEV))))
    assert y.shape == (1, 157, 16)
    assert maxdiff(y, z['logits']) <= 1e-3
    # Grid Pool frame indices from the model's OWN CDF == those of the reference's CDF (north_star: indices bit exact)
    from cfn_hip import ops
    from oracle import x3d_ref as R
    assert maxdiff(cdf, z['cdf']) <= 2e-6
    i_own, _ = ops.grid_time_index(cdf, 16)
    i_ref, _ = R.grid_sample_time_index(t(z['cdf']), 16)
    assert torch.equal(i_own.cpu()[:, :-1], i_ref[:, :-1]), (i_own.cpu(), i_ref)
    assert int((i_own.cpu()[:, -1] - i_ref[:, -1]).abs().max()) <= 1     # last knot: T-2 (w=1) == T-1 (w=0), see test_hip_ops


def test_coarse_train_fwd_bwd_vs_reference(op_route):
    z = load_golden('coarse_train')
    x, feat, fm, meta, depth = _coarse_inputs(110, 2, 16, 12)
    m = _coarse_model(depth, dropout=0.0)
    m.train(True)
    m.rw6.dropout.p = 0.0
    y = m([x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), 0, meta.to(DEV)])
    assert maxdiff(y, z['logits']) <= 1e-3
    from oracle import spec
    (y * spec.rand_input(120, tuple(y.shape)).to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    gn = json.loads(str(z['grad_norms']))
    # same conditioning caveat as test_fine_train_fwd_bwd_vs_reference (tests/test_oracle_golden.py documents the
    # measured 1-2 % sensitivity of this case to a 1e-6 input perturbation): norms 8 %, heads tight
    bad = []
    for k, ref in gn.items():
        assert named[k].grad is not None, k
        mine = float(named[k].grad.double().norm())
        # pool_1.conv3.bias: a uniform shift of the saliency logits -- its gradient is the sum of the per-knot
        # gradients, which cancel to ~1 % of their magnitude; the CPU oracle itself is 3.7 % off the reference there
        tol = 0.15 if k == 'pool_1.conv3.bias' else 8e-2
        if abs(mine - ref) > tol * max(ref, 1e-2):
            bad.append((k, mine, ref))
    assert not bad, bad[:10]
    for k in ('g_fc2_bias', 'g_rw6_at2_weight', 'g_mix5_conv_at2_weight'):
        name = [n for n in gn if ('g_' + n.replace('.', '_')) == k][0]
        assert relerr(thin(named[name].grad), z[k]) <= 1e-2, k


def test_coarse_run_to_run_reproducibility():
    """60 identical train-mode passes of the full Coarse-Fine net (2 clips x 16 frames + fine features): logits and EVERY parameter gradient
    bit-identical to the first pass (two passes with a tolerance do not catch a kernel that is wrong in 1-3 % of passes).
    Every reduction has a fixed order inside a workgroup; across workgroups fp32 partials meet in fp64 atomics (exact unless an addend is
    below 2^-29 of the running sum) or in the fixed-order reduce kernels."""
    x, feat, fm, meta, depth = _coarse_inputs(130, 2, 16, 12)
    m = _coarse_model(depth, dropout=0.0)
    m.train(True)
    m.rw6.dropout.p = 0.0
    from oracle import spec
    r = spec.rand_input(131, (2, 157, 16)).to(DEV)
    inp = [x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), 0, meta.to(DEV)]
    ref, bad = None, {}
    for _ in range(60):
        for p in m.parameters():
            p.grad = None
        y = m(inp)
        (y * r).sum().backward()
        cur = {'<logits>': y.detach().clone()}
        cur.update({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        if ref is None:
            ref = cur
            continue
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                bad[k] = bad.get(k, 0) + 1
    assert not bad, bad


def test_fine_run_to_run_bit_reproducibility():
    """150 identical train-mode passes of x3d_fine (2 x 16 frames): logits and EVERY parameter gradient bit-identical to the first pass.
    Regression test for a race found with tools/determinism_scan.py / tools/diag_wgrad_race.py in round 4: as compiled in round 3,
    `pws_wgrad_staged_kernel<2, 2, 2, ..>` (layer-2 conv3 weight gradient: 8 tiles on 8 waves x 2 tile slots, so four waves only staged
    operands and ran a phase ahead of the four that multiplied) wrote stale operand rows from lanes 48-63 of the staging-only waves in
    1-3 % of passes -- only in the model context (first launch of that variant in a backward pass), never in 2,400 isolated launches;
    the error (1e-5 .. 2e-3 of the gradient's norm) sat in two adjacent rows or columns of layer2.4.conv3.weight.grad.  With one tile per
    wave for shapes of <= 8 tiles: 0 of 2,000 passes."""
    import x3d_fine
    from oracle import spec
    net = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(net)
    net.to(DEV).train(True)
    x = spec.rand_input(5, (2, 3, 16, 224, 224)).to(DEV)
    ref, bad = None, {}
    for run in range(150):
        for p in net.parameters():
            p.grad = None
        y = net([x, None])
        if run == 0:
            r = spec.rand_input(777, tuple(y.shape)).to(DEV)
        (y * r).sum().backward()
        cur = {'<logits>': y.detach().clone()}
        cur.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        if ref is None:
            ref = cur
            continue
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                bad[k] = bad.get(k, 0) + 1
    assert not bad, bad


def test_x3d_xl_matches_oracle():
    """X3D-XL (72/162/306/630 channels, 5/10/25/15 blocks, SE width 40 at 630 channels: the squeeze-excite matrices do not fit
    in LDS and are read through L2) at odd plane sizes (160 -> 80, 40, 20, 10, 5): eval logits 1e-3, eval-mode gradients 2e-3"""
    import x3d_fine
    from oracle import spec, x3d_ref
    m = x3d_fine.generate_model('XL', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(m)
    m.to(DEV).eval()
    x = spec.rand_input(41, (1, 3, 4, 160, 160))
    y = m([x.to(DEV), None])
    r = spec.rand_input(42, tuple(y.shape))
    (y * r.to(DEV)).sum().backward()
    sd = spec.procedural_fill(spec.fine_keys('XL', 157, 1))
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    yo = x3d_ref.x3d_fine_forward(sd, x, 'XL', training=False)
    (yo * r).sum().backward()
    assert maxdiff(y, yo) <= 1e-3
    named = dict(m.named_parameters())
    for k in ('conv1_t.weight', 'layer1.0.conv2.weight', 'layer2.4.fc1.weight', 'layer3.24.conv3.weight', 'layer4.14.conv2.weight',
              'layer4.14.fc2.weight', 'conv5.weight', 'fc2.weight'):
        a, b = named[k].grad.cpu().double().flatten(), sd[k].grad.double().flatten()
        assert float((a - b).norm() / b.norm()) <= 2e-3, (k, float((a - b).norm() / b.norm()))
    m.train(True)                       # batch statistics + SE training path at 630 channels
    yt = m([spec.rand_input(43, (2, 3, 4, 160, 160)).to(DEV), None])
    yto = x3d_ref.x3d_fine_forward(spec.procedural_fill(spec.fine_keys('XL', 157, 1)), spec.rand_input(43, (2, 3, 4, 160, 160)), 'XL', training=True)
    assert maxdiff(yt, yto) <= 1e-3


def _live_device_tensors():
    import gc
    torch.cuda.synchronize()
    gc.collect()
    n = 0
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                n += 1
        except Exception:
            pass
    return n


@pytest.mark.gpu
def test_training_steps_do_not_leak_device_memory():
    """120 train steps of x3d_fine: the number of live device tensors after step 20 and after step 120 must agree (the byte count moves by a
    scratch-arena chunk either way).  (Round 5 found 30 tensors = 190 KB per step: the tail's backward left BN factors WITH their grad_fn in the
    TailLink object that the conv3 ctx holds -- a cycle through C++ autograd nodes, invisible to gc.collect().)"""
    import gc
    import x3d_fine
    from oracle import spec
    dev = torch.device('cuda:0')
    net = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.5)
    spec.fill_module_(net)
    net.to(dev).train(True)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    x = torch.randn(2, 3, 8, 96, 96, device=dev)
    lab = (torch.rand(2, 157, 8, device=dev) < 0.05).float()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(net([x, None]), lab)
        loss.backward()
        opt.step()

    for _ in range(20):
        step()
    m0 = _live_device_tensors()
    for _ in range(100):
        step()
    m1 = _live_device_tensors()
    assert m1 - m0 < 50, 'live device tensors grew by %d over 100 steps' % (m1 - m0)


@pytest.mark.gpu
def test_coarse_training_steps_do_not_leak_device_memory():
    """the same for the coarse stream's own train step (Grid Pool, fusion modules, gradient reducer)"""
    import gc
    import torch.optim as optim
    import train_coarse_fineFEAT as tc
    from cfn_hip import dist as cdist
    dev = torch.device('cuda:0')
    net = tc.build_model(dev, pretrained=None)
    optimizer = optim.SGD(tc.param_groups(net, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
    x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(2, 1, 16, seed=4321)))
    x = x[:, 0].contiguous().to(dev)
    labels, masks, fm, meta = labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
    feat = {k: v.to(dev) for k, v in feat.items()}
    net.train(True)
    reducer = cdist.GradReducer(net.parameters())

    try:
        for _ in range(20):
            tc.train_step(net, reducer, optimizer, x, labels, masks, feat, fm, meta)
        m0 = _live_device_tensors()
        for _ in range(60):
            tc.train_step(net, reducer, optimizer, x, labels, masks, feat, fm, meta)
        m1 = _live_device_tensors()
    finally:
        reducer.close()
    assert m1 - m0 < 50, 'live device tensors grew by %d over 60 steps' % (m1 - m0)


@pytest.mark.gpu
def test_two_threads_on_one_device_match_the_sequential_run():
    """the reference drives its model through nn.DataParallel: N Python threads of ONE process call forward concurrently and the autograd engine
    runs backward on worker threads.  Two threads, each with its own copy of x3d_fine on its own stream of the same device, run
    forward + backward concurrently; logits and every gradient must equal the sequential run (per-thread scratch arenas, per-(device, stream)
    workspaces, no unguarded global state)."""
    import threading
    import x3d_fine
    from oracle import spec
    dev = torch.device('cuda:0')

    def make():
        net = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
        spec.fill_module_(net)
        return net.to(dev).train(True)
    xs = [spec.rand_input(40 + i, (2, 3, 8, 64, 64)).to(dev) for i in range(2)]
    rs = [spec.rand_input(50 + i, (2, 157, 8)).to(dev) for i in range(2)]

    def run(net, x, r, out, stream=None):
        def body():
            for _ in range(3):
                net.zero_grad(set_to_none=True)
                y = net([x, None])
                (y * r).sum().backward()
            out.append((y.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
        if stream is None:
            body()
        else:
            with torch.cuda.stream(stream):
                body()
                stream.synchronize()
    ref = []
    for i in range(2):
        o = []
        run(make(), xs[i], rs[i], o)
        torch.cuda.synchronize()
        ref.append(o[0])
    nets = [make(), make()]
    outs = [[], []]
    errs = []

    def worker(i):
        try:
            run(nets[i], xs[i], rs[i], outs[i], torch.cuda.Stream(device=dev))
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(2):
        y, gr = outs[i][0]
        assert torch.allclose(y, ref[i][0], rtol=1e-5, atol=1e-6)
        for k, gv in gr.items():
            assert relerr(gv, ref[i][1][k]) <= 1e-5, k
