"""Pin the CPU oracle (oracle/x3d_ref.py) against vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only; runs in the build container and on the GPU box."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, golden_sd, t, maxdiff, relerr
from oracle import spec, x3d_ref as R

TOL = 2e-6


def test_state_dict_keys_match_reference():
    z = load_golden('state_keys')
    for field, mine in (('fine', spec.fine_keys('M', 157, 1)), ('coarse', spec.coarse_keys('M', 157, 1)),
                        ('fine_s2', spec.fine_keys('M', 157, 2))):
        ref = {k: tuple(s) for k, s in json.loads(str(z[field]))}
        assert set(ref) == set(mine), (set(ref) ^ set(mine))
        for k in ref:
            assert ref[k] == tuple(mine[k]), k
    assert len(spec.fine_keys('M', 157, 1)) == 820 and len(spec.coarse_keys('M', 157, 1)) == 918


@pytest.mark.parametrize('k', [5, 17, 65])
def test_interp1d(k):
    z = load_golden('interp1d_k%d' % k)
    y, ind = R.interp1d(t(z['x']), t(z['mid']), t(z['mid']))
    assert torch.equal(ind, t(z['ind']))
    assert torch.equal(y, t(z['ynew']))
    y2, ind2 = R.interp1d(t(z['x']), t(z['y2']), t(z['q2']))
    assert torch.equal(ind2, t(z['ind2']))
    assert maxdiff(y2, z['ynew2']) == 0.0


@pytest.mark.parametrize('tag', ['d4', 'd24'])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_gridpool(tag, mode):
    z = load_golden('gridpool_%s_%s' % (tag, mode))
    sd = {'pool.' + k: v for k, v in golden_sd(z).items()}
    x = spec.rand_input(int(z['seed']), tuple(int(v) for v in z['shape']))
    y, cdf = R.grid_pool(x, sd, 'pool', mode == 'train')
    assert maxdiff(cdf, z['cdf']) <= 1e-7
    i0, _ = R.grid_sample_time_index(t(z['cdf']), x.shape[2])
    assert torch.equal(i0, t(z['i0']))
    assert maxdiff(y, z['y']) <= 5e-6
    if mode == 'train':
        assert maxdiff(sd['pool.bn1.split_bn.running_mean'], z['rm1']) <= TOL
        assert maxdiff(sd['pool.bn2.split_bn.running_var'], z['rv2']) <= TOL


def test_gridsample_ulp_cases():
    z = load_golden('gridsample_ulp')
    y = R.grid_pool_resample(t(z['x']), t(z['cdf']))
    assert maxdiff(y, z['y']) == 0.0
    i0, _ = R.grid_sample_time_index(t(z['cdf']), z['x'].shape[2])
    assert torch.equal(i0, t(z['i0']))
    # all three last-knot cases are present: i0[-1] in {T-2, T-1}
    assert sorted(set(z['i0'][:, -1].tolist())) == [14, 15]


def test_gridunpool():
    z = load_golden('gridunpool')
    yl, inv, ind = R.grid_unpool(t(z['xl']), t(z['cdf']), True)
    assert maxdiff(yl, z['yl']) == 0.0
    up = F.interpolate(yl, (yl.shape[2] - 1) * 4, mode='linear', align_corners=True)
    assert maxdiff(up, z['yl_up']) == 0.0
    yf, _, _ = R.grid_unpool(t(z['xf']), t(z['cdf']), False)
    assert maxdiff(yf, z['yf']) == 0.0


def test_gaussian():
    z = load_golden('gaussian')
    g = R.gaussian(t(z['meta']), t(z['mask']), t(z['cdf']), int(z['T']))
    assert maxdiff(g, z['GX']) <= 1e-7


@pytest.mark.parametrize('name,hgt,mix,pool', [('rewight_h7_mix', 7, True, False), ('rewight_h7_nomix', 7, False, False),
                                               ('rewight_h14_mix', 14, True, False),
                                               ('rewight_h14_nomix', 14, False, False),
                                               ('rewight_pool', 7, False, True)])
def test_rewight(name, hgt, mix, pool):
    z = load_golden(name)
    sd = {'rw.' + k: v for k, v in golden_sd(z).items()}
    K = z['GX'].shape[2]
    b_, s_ = R.rewight(sd, 'rw', t(z['xf']), (z['xf'].shape[0], 0, K, hgt, hgt), t(z['mask']), t(z['GX']), mix, hgt,
                       pool=pool)
    assert maxdiff(b_, z['bias']) <= 2e-6
    assert maxdiff(s_, z['scale']) <= 2e-6


def test_gaussian_multicrop():
    """b2 = 2b: crop j of video i is offset by step_i * j (x3d_coarse.py:264-266)"""
    z = load_golden('gaussian_multicrop')
    g = R.gaussian(t(z['meta']), t(z['mask']), t(z['cdf']), int(z['T']))
    assert g.shape == (4, 20, 9) and maxdiff(g, z['GX']) <= 1e-7


@pytest.mark.parametrize('mix', [True, False])
def test_rewight_multicrop(mix):
    z = load_golden('rewight_multicrop_%s' % ('mix' if mix else 'nomix'))
    sd = {'rw.' + k: v for k, v in golden_sd(z).items()}
    b2, K = z['GX'].shape[0], z['GX'].shape[2]
    b_, s_ = R.rewight(sd, 'rw', t(z['xf']), (b2, 0, K, 14, 14), t(z['mask']), t(z['GX']), mix, 14)
    assert b_.shape[0] == 2 * z['xf'].shape[0]
    assert maxdiff(b_, z['bias']) <= 2e-6 and maxdiff(s_, z['scale']) <= 2e-6


def test_loss_multicrop():
    z = load_golden('loss_multicrop')
    for ac in (1, 0):
        cls, loc, probs = R.detection_loss(t(z['logits']), t(z['labels']), t(z['masks']), bool(ac), crops=int(z['crops']))
        assert abs(float(cls) - float(z['cls_%d' % ac])) <= 1e-7 and abs(float(loc) - float(z['loc_%d' % ac])) <= 1e-7
        assert maxdiff(probs[:, ::13], z['probs_%d' % ac]) <= 1e-7


@pytest.mark.parametrize('li,h', [(0, 14), (3, 7)])
def test_mixing(li, h):
    z = load_golden('mixing_l%d' % li)
    sd = {'mix.' + k: v for k, v in golden_sd(z).items()}
    B, K = 1, int(z['K'])
    chans = (24, 48, 96, 192)
    bias, scale = [], []
    for j, (c, hh) in enumerate(zip(chans, (56, 28, 14, 7))):
        up = lambda v: F.adaptive_max_pool2d(v.view(B, c * K, 7, 7), (hh, hh)).view(B, c, K, hh, hh)
        bias.append(up(spec.rand_input(60 + j, (B, c, K, 7, 7))))
        scale.append(up(spec.rand_input(70 + j, (B, c, K, 7, 7))))
    c_, m_ = R.mixing(sd, 'mix', (B, chans[li], K, h, h), bias, scale)
    assert maxdiff(c_, z['c']) <= 5e-6
    assert maxdiff(m_, z['m']) <= 2e-6


@pytest.mark.parametrize('S', [1, 2])
def test_subbn(S):
    z = load_golden('subbn_s%d' % S)
    sd = {'bn.' + k: v for k, v in golden_sd(z).items()}
    x1 = spec.rand_input(81, (4, 6, 3, 5, 5)) * 1.7 + 0.3
    x2 = spec.rand_input(82, (4, 6, 3, 5, 5)) * 0.6 - 0.2
    assert maxdiff(R.sub_bn(x1, sd, 'bn', True, S), z['y1']) <= TOL
    assert maxdiff(R.sub_bn(x2, sd, 'bn', True, S), z['y2']) <= TOL
    assert maxdiff(sd['bn.split_bn.running_mean'], z['split_rm']) <= TOL
    assert maxdiff(sd['bn.split_bn.running_var'], z['split_rv']) <= TOL
    R.aggregate_bn_stats(sd, S)
    assert maxdiff(sd['bn.bn.running_mean'], z['rm']) <= TOL
    assert maxdiff(sd['bn.bn.running_var'], z['rv']) <= TOL
    assert maxdiff(R.sub_bn(x1, sd, 'bn', False, S), z['y3']) <= TOL


# (tag, index, stride, cin, input shape): as tests/golden/make_golden.py BOTTLENECK_CASES (l3_* / l4_*: layer-3 / 4 widths at 14x14 / 7x7)
BOTTLENECKS = [('even_s1', 0, 1, 24, (2, 24, 4, 8, 8)), ('odd_s1', 1, 1, 24, (2, 24, 4, 8, 8)), ('even_s2', 0, 2, 24, (2, 24, 4, 8, 8)),
               ('odd_s2', 1, 2, 48, (2, 48, 4, 8, 8)), ('l3_even_s2', 0, 2, 48, (2, 48, 2, 28, 28)), ('l3_odd_s1', 1, 1, 96, (2, 96, 2, 14, 14)),
               ('l4_even_s2', 0, 2, 96, (2, 96, 2, 14, 14)), ('l4_odd_s1', 1, 1, 192, (2, 192, 2, 7, 7))]


@pytest.mark.parametrize('tag,index,stride,cin,shape', BOTTLENECKS)
def test_bottleneck_fwd_bwd(tag, index, stride, cin, shape):
    z = load_golden('bottleneck_' + tag)
    sd = {'b.' + k: v for k, v in golden_sd(z).items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    x = F.relu(spec.rand_input(91, shape)).requires_grad_(True)
    y = R.bottleneck(x, sd, 'b', stride, index, True, 1)
    assert maxdiff(y, z['y']) <= 5e-6
    (y * spec.rand_input(92, tuple(y.shape))).sum().backward()
    assert relerr(x.grad, z['gx']) <= 2e-5
    for k in z:
        if k.startswith('g_'):
            name = 'b.' + k[2:].replace('_weight', '.weight').replace('_bias', '.bias').replace('downsample_', 'downsample.')
            assert relerr(sd[name].grad, z[k]) <= 5e-5, k
    assert maxdiff(sd['b.bn2.split_bn.running_mean'], z['bn2_rm']) <= TOL
    assert maxdiff(sd['b.bn2.split_bn.running_var'], z['bn2_rv']) <= TOL


def test_loss_and_ap():
    z = load_golden('loss_ap')
    for ac in (1, 0):
        cls, loc, _ = R.detection_loss(t(z['logits']), t(z['labels']), t(z['masks']), bool(ac))
        assert abs(float(cls) - float(z['cls_%d' % ac])) <= 1e-7
        assert abs(float(loc) - float(z['loc_%d' % ac])) <= 1e-7
    ap = R.average_precision(z['ap_scores'], z['ap_targets'])
    assert np.abs(ap - z['ap']).max() <= 1e-6
    # known answer recorded in SURVEY.md Appendix A
    assert np.allclose(ap, [0.3568, 0.2009, 0.2818, 0.2449, 0.4016], atol=5e-4)


def test_fine_cfg1_logits():
    """BASELINE.json configs[0]: X3D-S eval forward, 1x3x13x160x160."""
    z = load_golden('fine_cfg1')
    sd = golden_sd(z)
    with torch.no_grad():
        y = R.x3d_fine_forward(sd, spec.rand_input(0, (1, 3, 13, 160, 160)), 'S', training=False)
    assert y.shape == (1, 157, 13)
    assert maxdiff(y, z['logits']) <= 2e-5


def test_fine_tower():
    z = load_golden('fine_tower')
    sd = spec.procedural_fill(spec.fine_keys('M', 157, 1))
    with torch.no_grad():
        f = R.x3d_fine_forward(sd, spec.rand_input(1, (1, 3, 6, 64, 64)), 'M', training=False, global_tower=True)
    for k in ('layer1', 'layer2', 'layer3', 'layer4', 'conv5'):
        assert maxdiff(f[k], z[k]) <= 1e-5, k


def thin(v, limit=20000):
    return v if v.numel() <= limit else v.flatten()[::37]


def test_fine_train_fwd_bwd():
    z = load_golden('fine_train')
    sd = spec.procedural_fill(spec.fine_keys('M', 157, 1))
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    y = R.x3d_fine_forward(sd, spec.rand_input(2, (2, 3, 8, 64, 64)), 'M', training=True)
    assert maxdiff(y, z['logits']) <= 1e-4
    (y * spec.rand_input(3, tuple(y.shape))).sum().backward()
    gn = json.loads(str(z['grad_norms']))
    for k, ref in gn.items():
        mine = float(sd[k].grad.double().norm())
        assert abs(mine - ref) <= 2e-3 * max(ref, 1e-3), (k, mine, ref)
    for k in z:
        if k.startswith('g_'):
            name = [n for n in gn if ('g_' + n.replace('.', '_')) == k][0]
            assert relerr(thin(sd[name].grad), z[k]) <= 2e-3, k
    assert maxdiff(sd['layer2.0.bn2.split_bn.running_mean'], z['bn_rm']) <= 1e-5


def _coarse_inputs(seed, B, T, Tf):
    x = spec.rand_input(seed, (B, 3, T, 224, 224))
    depth = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}
    feat = {k: spec.rand_input(seed + 1 + i, (B, c, Tf, 7, 7), nonneg=True) for i, (k, c) in enumerate(depth.items())}
    fm = torch.ones(B, Tf)
    meta = torch.zeros(B, 4, dtype=torch.int64)
    for b in range(B):
        valid = Tf - 3 * b
        fm[b, valid:] = 0
        meta[b] = torch.tensor([b * 2, T, valid, 1])
    return x, feat, fm, meta


def test_coarse_eval_logits():
    z = load_golden('coarse_eval')
    sd = golden_sd(z)
    x, feat, fm, meta = _coarse_inputs(100, 1, 16, 12)
    with torch.no_grad():
        y, aux = R.x3d_coarse_forward(sd, [x, feat, fm, 0, meta], 'M', training=False, return_aux=True)
    assert maxdiff(aux['cdf'], z['cdf']) <= 1e-6
    assert maxdiff(y, z['logits']) <= 5e-5


def test_coarse_train_fwd_bwd():
    z = load_golden('coarse_train')
    sd = spec.procedural_fill(spec.coarse_keys('M', 157, 1))
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    x, feat, fm, meta = _coarse_inputs(110, 2, 16, 12)
    y = R.x3d_coarse_forward(sd, [x, feat, fm, 0, meta], 'M', training=True)
    assert maxdiff(y, z['logits']) <= 2e-4
    (y * spec.rand_input(120, tuple(y.shape))).sum().backward()
    gn = json.loads(str(z['grad_norms']))
    for k, ref in gn.items():
        mine = float(sd[k].grad.double().norm())
        # floor: biases feeding train-mode BN have zero true grad.  5e-2: measured -- scaling the input clip by
        # (1+1e-6) moves the trunk gradients of this tiny (B=2, K=5, 7x7 at layer4) case by 1-2 % (ReLU / max
        # kinks in front of batch-stat BN), i.e. the case is ill-conditioned in fp32; heads (fc2, rw6, mix5) agree
        # to 1e-5.  Tight backward checks are the module-level fixtures (bottleneck_* at 5e-5).
        assert abs(mine - ref) <= 5e-2 * max(ref, 1e-2), (k, mine, ref)
    for k in z:
        if k.startswith('g_'):
            name = [n for n in gn if ('g_' + n.replace('.', '_')) == k][0]
            assert relerr(thin(sd[name].grad), z[k]) <= 5e-2, k
    for k in ('g_fc2_bias', 'g_rw6_at2_weight', 'g_mix5_conv_at2_weight'):
        name = [n for n in gn if ('g_' + n.replace('.', '_')) == k][0]
        assert relerr(thin(sd[name].grad), z[k]) <= 1e-4, k


def test_product_apmeter_matches_reference():
    """coarse-fine-networks_amd/apmeter.py (host-side metric) against the reference's APMeter output"""
    from apmeter import APMeter
    z = load_golden('loss_ap')
    m = APMeter()
    for i in range(3):
        m.add(z['ap_scores'][20 * i:20 * i + 20], z['ap_targets'][20 * i:20 * i + 20])
    assert np.abs(m.value().numpy() - z['ap']).max() <= 1e-6
    m.reset()
    assert m.value() == 0
