#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py            # needs /root/reference (read-only mount)

The reference has no tests or golden vectors of its own, so parity of
oracle/ (and through it of the HIP path) is pinned by the vectors this script captures from
the reference modules imported read-only.  Nothing of the reference's source is copied: only
inputs (or their seeds) and the outputs the reference computed are stored.  Weights are the
procedural by-key-name fill of oracle/spec.py, so no weight blobs are stored either.

The GPU box has no /root/reference; it only ever reads the committed .npz files.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('CFN_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

torch.Tensor.cuda = lambda self, *a, **k: self   # the reference hard-codes .cuda() (x3d_coarse.py:265,...)
torch.set_num_threads(8)

import x3d_fine as ref_fine            # noqa: E402  (reference)
import x3d_coarse as ref_coarse        # noqa: E402
import interp1d as ref_interp          # noqa: E402
import apmeter as ref_apmeter          # noqa: E402

from oracle import spec                # noqa: E402


def keys_json(module):
    return json.dumps([[k, list(v.shape)] for k, v in module.state_dict().items()])


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in out.items()})


def thin(v, limit=20000):
    """big gradient tensors are stored as a strided subsample of the flattened tensor
    (stride 37; the tests subsample the oracle / HIP result identically)"""
    return v if v.numel() <= limit else v.flatten()[::37]


def ref_ind(x, xnew):
    """The index tensor Interp1d builds internally (interp1d.py:100-110) -- it is not
    returned by the reference, so recompute it with the identical torch calls."""
    ind = torch.searchsorted(x.contiguous(), xnew.contiguous()) - 1
    return torch.clamp(ind, 0, x.shape[1] - 2)


def make_cdf(rs, b, k, last):
    """monotone CDF rows with cdf[0]=0 and a prescribed last knot (ulp cases of SURVEY 7)."""
    p = rs.uniform(0.2, 1.0, size=(b, k - 1)).astype(np.float32)
    c = np.cumsum(p / p.sum(1, keepdims=True), axis=1, dtype=np.float64).astype(np.float32)
    out = np.zeros((b, k), np.float32)
    out[:, 1:] = c
    for i, l in enumerate(last):
        out[i, -1] = l
    return out


def case_interp1d():
    rs = np.random.RandomState(11)
    for k in (5, 17, 65):
        lasts = [np.float32(1) - np.float32(2 ** -24), np.float32(1), np.float32(1) + np.float32(2 ** -23),
                 np.float32(1)]
        x = torch.from_numpy(make_cdf(rs, 4, k, lasts))
        mid = (torch.arange(k).to(torch.float32) / (k - 1.)).view(1, -1).repeat(4, 1)
        ynew = ref_interp.Interp1d()(x, mid, mid, None)
        # generic y / xnew too (not only the unpool usage)
        y2 = torch.from_numpy(rs.standard_normal((4, k)).astype(np.float32))
        q2 = torch.from_numpy(np.sort(rs.uniform(-0.1, 1.1, size=(4, 23)).astype(np.float32), axis=1))
        ynew2 = ref_interp.Interp1d()(x, y2, q2, None)
        save('interp1d_k%d' % k, x=x, mid=mid, ynew=ynew, ind=ref_ind(x, mid),
             y2=y2, q2=q2, ynew2=ynew2, ind2=ref_ind(x, q2))


def case_gridpool():
    for tag, depth, shape, seed in (('d4', 4, (2, 4, 16, 8, 8), 21), ('d24', 24, (1, 24, 64, 14, 14), 22)):
        for training in (False, True):
            m = ref_coarse.GridPoolLayer(4, depth)
            spec.fill_module_(m)
            m.train(training)
            x = spec.rand_input(seed, shape)
            with torch.no_grad():
                y, cdf = m(x)
            T = shape[2]
            it = (((cdf - 0.5) * 2 + 1) / 2) * (T - 1)
            save('gridpool_%s_%s' % (tag, 'train' if training else 'eval'), keys=keys_json(m),
                 seed=seed, shape=np.array(shape), y=y, cdf=cdf, i0=torch.floor(it).to(torch.int32),
                 rm1=m.bn1.split_bn.running_mean, rv2=m.bn2.split_bn.running_var)
    # resampler alone on hand-made CDFs incl. the three last-knot ulp cases
    rs = np.random.RandomState(23)
    T, K = 16, 5
    cdf = torch.from_numpy(make_cdf(rs, 3, K, [np.float32(1) - np.float32(2 ** -24), np.float32(1),
                                               np.float32(1) + np.float32(2 ** -23)]))
    x = spec.rand_input(24, (3, 2, T, 6, 5))
    gx = (cdf - 0.5) * 2
    gh = (torch.arange(6).to(torch.float32) / 5 - 0.5) * 2
    gw = (torch.arange(5).to(torch.float32) / 4 - 0.5) * 2
    grid = torch.meshgrid([gx.view(-1), gh, gw])
    grid = torch.stack((grid[2], grid[1], grid[0]), dim=-1).view(3, K, 6, 5, 3)
    y = F.grid_sample(x, grid, align_corners=True)
    it = ((gx + 1) / 2) * (T - 1)
    save('gridsample_ulp', x=x, cdf=cdf, y=y, i0=torch.floor(it).to(torch.int32))


def case_gridunpool():
    rs = np.random.RandomState(31)
    cdf = torch.from_numpy(make_cdf(rs, 2, 17, [np.float32(1), np.float32(1) - np.float32(2 ** -24)]))
    xl = spec.rand_input(32, (2, 7, 17))
    yl = ref_coarse.GridUnpool([xl, cdf, True])
    yl_up = F.interpolate(yl, (yl.shape[2] - 1) * 4, mode='linear', align_corners=True)
    xf = spec.rand_input(33, (2, 3, 17, 6, 6))
    yf = ref_coarse.GridUnpool([xf, cdf, False])
    save('gridunpool', cdf=cdf, xl=xl, yl=yl, yl_up=yl_up, xf=xf, yf=yf)


def case_gaussian():
    rs = np.random.RandomState(41)
    B, Tf, K, T = 3, 40, 17, 64
    cdf = torch.from_numpy(make_cdf(rs, B, K, [np.float32(1)] * B))
    mask = torch.ones(B, Tf)
    mask[1, 25:] = 0
    meta = torch.tensor([[0, 64, 40, 1], [7, 64, 25, 1], [13, 64, 40, 1]], dtype=torch.int64)
    g = ref_coarse.Gaussian(ratio=1)([meta, mask, cdf, T])
    save('gaussian', cdf=cdf, mask=mask, meta=meta, T=np.array(T), GX=g)
    return cdf, mask, meta, g


def case_rewight():
    rs = np.random.RandomState(51)
    B, C, Tf, K = 2, 8, 12, 5
    cdf = torch.from_numpy(make_cdf(rs, B, K, [np.float32(1)] * B))
    mask = torch.ones(B, Tf)
    mask[1, 9:] = 0
    meta = torch.tensor([[0, 16, 12, 1], [2, 16, 9, 1]], dtype=torch.int64)
    GX = ref_coarse.Gaussian(ratio=1)([meta, mask, cdf, 16])
    for hgt in (7, 14):
        for is_mix in (True, False):
            m = ref_coarse.RewightLayer(channels=6, g_channels=6, depth=C, height=hgt)
            spec.fill_module_(m)
            m.eval()
            xf = spec.rand_input(52, (B, C, Tf, 7, 7), nonneg=True)
            lx = torch.zeros(B, 6, K, hgt, hgt)
            with torch.no_grad():
                b_, s_ = m([xf, lx, mask, None, 0, GX, is_mix])
            save('rewight_h%d_%s' % (hgt, 'mix' if is_mix else 'nomix'), keys=keys_json(m), xf=xf, mask=mask,
                 GX=GX, bias=b_, scale=s_)
    # pooled variant (rw6)
    m = ref_coarse.RewightLayer(channels=5, g_channels=5, depth=C, height=7, pool=True)
    spec.fill_module_(m)
    m.eval()
    xf = spec.rand_input(53, (B, C, Tf, 7, 7), nonneg=True)
    with torch.no_grad():
        b_, s_ = m([xf, torch.zeros(B, 5, K, 1, 1), mask, None, 0, GX, False])
    save('rewight_pool', keys=keys_json(m), xf=xf, mask=mask, GX=GX, bias=b_, scale=s_)


def case_mixing():
    B, K = 1, 3
    chans = (24, 48, 96, 192)
    for li, h in ((0, 14), (3, 7)):
        m = ref_coarse.MixingLayer(depth=chans[li], learned=True, index=li)
        spec.fill_module_(m)
        m.eval()
        # fusion outputs are 7x7-block-constant upsamples (adaptive_max_pool2d of 7x7 inputs)
        bias, scale = [], []
        for j, (c, hh) in enumerate(zip(chans, (56, 28, 14, 7))):
            b7 = spec.rand_input(60 + j, (B, c, K, 7, 7))
            s7 = spec.rand_input(70 + j, (B, c, K, 7, 7))
            up = lambda v: F.adaptive_max_pool2d(v.view(B, c * K, 7, 7), (hh, hh)).view(B, c, K, hh, hh)
            bias.append(up(b7))
            scale.append(up(s7))
        x = torch.zeros(B, chans[li], K, h, h)
        with torch.no_grad():
            c_, m_ = m([x, bias, scale])
        save('mixing_l%d' % li, keys=keys_json(m), c=c_, m=m_, h=np.array(h), K=np.array(K))


def case_subbn():
    for S in (1, 2):
        m = ref_fine.SubBatchNorm3d(num_splits=S, num_features=6, affine=True)
        spec.fill_module_(m)
        x1 = spec.rand_input(81, (4, 6, 3, 5, 5)) * 1.7 + 0.3
        x2 = spec.rand_input(82, (4, 6, 3, 5, 5)) * 0.6 - 0.2
        m.train(True)
        y1 = m(x1)
        y2 = m(x2)
        m.aggregate_stats()
        m.train(False)
        y3 = m(x1)
        save('subbn_s%d' % S, keys=keys_json(m), y1=y1, y2=y2, y3=y3,
             split_rm=m.split_bn.running_mean, split_rv=m.split_bn.running_var,
             rm=m.bn.running_mean, rv=m.bn.running_var)


# (tag, index, stride, cin, planes, input shape).  The first four are X3D layer-1 / 2 widths on 8x8 planes; the `l3_*` / `l4_*`
# cases are the layer-3 / 4 widths (216- and 432-channel conv2 / SE, the split-bf16 and fp32-MFMA pointwise kernels, the 14x14 /
# 7x7 depthwise kernels) at their real plane sizes: train-mode forward + backward pinned to the REFERENCE, not only per op.
BOTTLENECK_CASES = (('even_s1', 0, 1, 24, (54, 24), (2, 24, 4, 8, 8)), ('odd_s1', 1, 1, 24, (54, 24), (2, 24, 4, 8, 8)),
                    ('even_s2', 0, 2, 24, (54, 48), (2, 24, 4, 8, 8)), ('odd_s2', 1, 2, 48, (108, 48), (2, 48, 4, 8, 8)),
                    ('l3_even_s2', 0, 2, 48, (216, 96), (2, 48, 2, 28, 28)), ('l3_odd_s1', 1, 1, 96, (216, 96), (2, 96, 2, 14, 14)),
                    ('l4_even_s2', 0, 2, 96, (432, 192), (2, 96, 2, 14, 14)), ('l4_odd_s1', 1, 1, 192, (432, 192), (2, 192, 2, 7, 7)))


def case_bottleneck():
    for tag, index, stride, cin, planes, shape in BOTTLENECK_CASES:
        ds = None
        if stride != 1 or cin != planes[1]:
            ds = torch.nn.Sequential(ref_fine.conv1x1x1(cin, planes[1], stride),
                                     ref_fine.SubBatchNorm3d(num_splits=1, num_features=planes[1], affine=True))
        m = ref_fine.Bottleneck(cin, planes, stride, ds, index=index, base_bn_splits=1)
        spec.fill_module_(m)
        m.train(True)
        x = F.relu(spec.rand_input(91, shape)).requires_grad_(True)
        y = m(x)
        r = spec.rand_input(92, tuple(y.shape))
        (y * r).sum().backward()
        grads = {('g_' + n.replace('.', '_')): p.grad for n, p in m.named_parameters()}
        save('bottleneck_' + tag, keys=keys_json(m), y=y, gx=x.grad, bn2_rm=m.bn2.split_bn.running_mean,
             bn2_rv=m.bn2.split_bn.running_var, **grads)


def case_fine():
    # cfg1: X3D-S eval forward on 1x3x13x160x160 (BASELINE.json configs[0])
    m = ref_fine.generate_model('S', n_classes=400, task='loc', base_bn_splits=1)
    m.replace_logits(157)
    spec.fill_module_(m)
    m.eval()
    with torch.no_grad():
        y = m([spec.rand_input(0, (1, 3, 13, 160, 160)), None])
    save('fine_cfg1', keys=keys_json(m), logits=y)
    # feature tower
    m2 = ref_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, global_tower=True)
    spec.fill_module_(m2)
    m2.eval()
    with torch.no_grad():
        f, _ = m2([spec.rand_input(1, (1, 3, 6, 64, 64)), None])
    save('fine_tower', **{k: v for k, v in f.items()})
    # train-mode fwd+bwd (dropout off: RNG streams are device specific), X3D-M, 2x3x8x64x64
    m3 = ref_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(m3)
    m3.train(True)
    x = spec.rand_input(2, (2, 3, 8, 64, 64))
    y = m3([x, None])
    r = spec.rand_input(3, tuple(y.shape))
    (y * r).sum().backward()
    pick = ['conv1_s.weight', 'conv1_t.weight', 'bn1.weight', 'layer1.0.conv1.weight', 'layer1.0.conv2.weight',
            'layer1.0.fc1.weight', 'layer1.0.downsample.0.weight', 'layer2.1.bn2.bias', 'layer3.4.conv3.weight',
            'layer4.6.conv2.weight', 'conv5.weight', 'fc1.weight', 'fc2.weight', 'fc2.bias']
    named = dict(m3.named_parameters())
    gn = {k: float(p.grad.double().norm()) for k, p in named.items()}
    save('fine_train', logits=y, grad_norms=json.dumps(gn),
         bn_rm=m3.layer2[0].bn2.split_bn.running_mean, bn_rv=m3.layer2[0].bn2.split_bn.running_var,
         **{('g_' + k.replace('.', '_')): thin(named[k].grad) for k in pick})


def coarse_inputs(seed, B, T, Tf):
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


def case_coarse():
    B, T, Tf = 1, 16, 12
    x, feat, fm, meta, depth = coarse_inputs(100, B, T, Tf)
    m = ref_coarse.generate_model('M', n_classes=400, feat_depth=depth, task='loc', dropout=0.0, base_bn_splits=1,
                                  learnedMixing=True, isMixing=True, t_pool='grid')
    m.replace_logits(157)
    spec.fill_module_(m)
    m.eval()
    with torch.no_grad():
        y = m([x, feat, fm, 0, meta])
        # capture the CDF the model used
        h = m.relu(m.bn1(m.conv1_t(m.conv1_s(x))))
        h = m.layer1(h)
        _, cdf = m.pool_1(h)
    save('coarse_eval', keys=keys_json(m), logits=y, cdf=cdf)
    # train fwd+bwd, B=2 (second sample has a shorter valid fine length and a non-zero start)
    B = 2
    x, feat, fm, meta, depth = coarse_inputs(110, B, T, Tf)
    m.train(True)
    m.rw6.dropout.p = 0.0
    y = m([x, feat, fm, 0, meta])
    r = spec.rand_input(120, tuple(y.shape))
    (y * r).sum().backward()
    named = dict(m.named_parameters())
    pick = ['pool_1.conv1.weight', 'pool_1.conv3.weight', 'pool_1.conv3.bias', 'conv1_t.weight', 'layer1.2.conv2.weight',
            'layer2.0.conv1.weight', 'rw2.at1.weight', 'rw2.fc2.weight', 'rw4.fc4.bias', 'rw6.at2.weight',
            'mix2.conv_at.weight', 'mix5.conv_at2.weight', 'layer4.0.conv2.weight', 'fc2.bias']
    gn = {k: float(p.grad.double().norm()) for k, p in named.items() if p.grad is not None}
    save('coarse_train', logits=y, grad_norms=json.dumps(gn),
         **{('g_' + k.replace('.', '_')): thin(named[k].grad) for k in pick})


def case_loss_ap():
    lg = spec.rand_input(130, (2, 157, 16))
    labels = torch.from_numpy((np.random.RandomState(131).uniform(size=(2, 157, 40)) < 0.05).astype(np.float32))
    masks = torch.ones(2, 40)
    masks[1, 30:] = 0
    out = {}
    for ac in (True, False):
        pl = F.interpolate(lg, 40, mode='linear', align_corners=ac)
        probs = torch.sigmoid(pl) * masks.unsqueeze(1)
        cls = torch.nn.BCELoss(reduction='mean')(torch.max(probs, dim=2)[0], torch.max(labels, dim=2)[0])
        loc = torch.nn.BCELoss(reduction='sum')(probs, labels) / (torch.sum(masks) * labels.shape[1])
        out['cls_%d' % ac] = cls
        out['loc_%d' % ac] = loc
    np.random.seed(0)
    apm = ref_apmeter.APMeter()
    ss, ts = [], []
    for _ in range(3):
        s = np.random.rand(20, 5).astype(np.float32)
        t = (np.random.rand(20, 5) > 0.7).astype(np.float32)
        apm.add(s, t)
        ss.append(s)
        ts.append(t)
    save('loss_ap', logits=lg, labels=labels, masks=masks, ap=apm.value(), ap_scores=np.concatenate(ss),
         ap_targets=np.concatenate(ts), **out)


def case_keys():
    m = ref_fine.generate_model('M', n_classes=400, task='loc', base_bn_splits=1)
    m.replace_logits(157)
    depth = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}
    c = ref_coarse.generate_model('M', n_classes=400, feat_depth=depth, task='loc', base_bn_splits=1,
                                  learnedMixing=True, isMixing=True, t_pool='grid')
    c.replace_logits(157)
    m2 = ref_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=2)
    save('state_keys', fine=keys_json(m), coarse=keys_json(c), fine_s2=keys_json(m2))


def case_multicrop():
    """validation-time multi-crop (b2 = n*b): Gaussian offsets `st` by step*crop (x3d_coarse.py:264-266), RewightLayer
    repeats the fine features / mask per crop (:209-211), the loss takes the max over the crops of a video
    (train_fine.py:204-207)."""
    rs = np.random.RandomState(141)
    B, n, Tf, K, T = 2, 2, 20, 9, 32
    cdf = torch.from_numpy(make_cdf(rs, B * n, K, [np.float32(1)] * (B * n)))
    mask = torch.ones(B, Tf)
    mask[1, 14:] = 0
    meta = torch.tensor([[0, 32, 20, 3], [5, 32, 14, 2]], dtype=torch.int64)      # step (col 3) != 1: the crop offset shows
    GX = ref_coarse.Gaussian(ratio=1)([meta, mask, cdf, T])
    save('gaussian_multicrop', cdf=cdf, mask=mask, meta=meta, T=np.array(T), GX=GX)
    C = 8
    for is_mix in (True, False):
        m = ref_coarse.RewightLayer(channels=6, g_channels=6, depth=C, height=14)
        spec.fill_module_(m)
        m.eval()
        xf = spec.rand_input(142, (B, C, Tf, 7, 7), nonneg=True)
        lx = torch.zeros(B * n, 6, K, 14, 14)
        with torch.no_grad():
            b_, s_ = m([xf, lx, mask, None, 0, GX, is_mix])
        save('rewight_multicrop_%s' % ('mix' if is_mix else 'nomix'), keys=keys_json(m), xf=xf, mask=mask, GX=GX,
             bias=b_, scale=s_)
    # loss with n = 3 crops per video, both align_corners conventions (the val branch of both training scripts)
    b, ncrop, tl = 2, 3, 40
    lg = spec.rand_input(143, (b * ncrop, 157, 16))
    labels = torch.from_numpy((np.random.RandomState(144).uniform(size=(b, 157, tl)) < 0.05).astype(np.float32))
    masks = torch.ones(b, tl)
    masks[0, 33:] = 0
    out = {}
    for ac in (True, False):
        pl = F.interpolate(lg, tl, mode='linear', align_corners=ac).view(b, ncrop, -1, tl)
        probs = torch.max(torch.sigmoid(pl), dim=1)[0] * masks.unsqueeze(1)
        out['cls_%d' % ac] = torch.nn.BCELoss(reduction='mean')(torch.max(probs, dim=2)[0], torch.max(labels, dim=2)[0])
        out['loc_%d' % ac] = torch.nn.BCELoss(reduction='sum')(probs, labels) / (torch.sum(masks) * labels.shape[1])
        out['probs_%d' % ac] = probs[:, ::13]
    save('loss_multicrop', logits=lg, labels=labels, masks=masks, crops=np.array(ncrop), **out)


def case_collate():
    """the two `mt_collate_fn` batch builders (charades_fine.py:201-224, charades_coarse_fineFEAT.py:208-252) on ragged
    samples.  The dataset modules import h5py / torchvision / cv2 at module level (absent here, and unused by the collate
    functions): empty placeholder modules satisfy those imports; the functions themselves run unmodified."""
    import types
    for name in ('h5py', 'torchvision', 'cv2', 'accimage'):
        sys.modules.setdefault(name, types.ModuleType(name))
    import charades_fine as ref_cf
    import charades_coarse_fineFEAT as ref_cc
    rs = np.random.RandomState(151)
    lens = [(6, 60), (9, 90), (4, 37)]
    fine_batch = [(rs.standard_normal((2, 3, t, 4, 5)).astype(np.float32),
                   (rs.uniform(size=(7, tl)) < 0.3).astype(np.float32), 'vid%d' % i) for i, (t, tl) in enumerate(lens)]
    clips, label, mask, vids = ref_cf.mt_collate_fn(fine_batch)
    save('collate_fine', clips=clips, label=label, mask=mask, vids=json.dumps(list(vids)),
         **{'in%d_%s' % (i, k): v for i, b in enumerate(fine_batch) for k, v in (('clips', b[0]), ('label', b[1]))})
    chans = {'layer1': 3, 'conv5': 5}
    flens = [100, 140, 128]       # one sample beyond the 128-frame cap
    coarse_batch = []
    for i, ((t, tl), tf) in enumerate(zip(lens, flens)):
        feat = {k: np.abs(rs.standard_normal((c, tf, 2, 2))).astype(np.float32) for k, c in chans.items()}
        coarse_batch.append((rs.standard_normal((1, 3, t, 4, 5)).astype(np.float32),
                             (rs.uniform(size=(7, tl)) < 0.3).astype(np.float32), feat,
                             np.array([i, t, tf, 1]), 'vid%d' % i, 10.5 + i))
    clips, label, mask, feat, fmask, meta, vids, dur = ref_cc.mt_collate_fn(coarse_batch)
    ins = {}
    for i, b in enumerate(coarse_batch):
        ins['in%d_clips' % i], ins['in%d_label' % i], ins['in%d_meta' % i] = b[0], b[1], b[3]
        for k in chans:
            ins['in%d_feat_%s' % (i, k)] = b[2][k]
    save('collate_coarse', clips=clips, label=label, mask=mask, fmask=fmask, meta=meta, dur=dur,
         vids=json.dumps(list(vids)), **{'feat_' + k: v for k, v in feat.items()}, **ins)


def case_options():
    """constructor options the reference scripts never set but the reference implements: temporal down-sampling
    (t_downsample, x3d_fine.py:93,104), the parameter-free shortcut 'A' (:266-275, which only fits t_downsample=True), and
    the fixed temporal poolings of the coarse stream t_pool = avg | max | stride | None (x3d_coarse.py:489-492, :640-660)."""
    x = spec.rand_input(160, (1, 3, 16, 64, 64))
    for tag, kw in (('tdown', dict(t_downsample=True)), ('tdown_A', dict(t_downsample=True, shortcut_type='A'))):
        m = ref_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0, **kw)
        spec.fill_module_(m)
        m.eval()
        with torch.no_grad():
            y = m([x, None])
        m.train(True)
        xt = spec.rand_input(161, (2, 3, 8, 64, 64))
        yt = m([xt, None])
        r = spec.rand_input(162, tuple(yt.shape))
        (yt * r).sum().backward()
        named = dict(m.named_parameters())
        pick = ['fc2.weight', 'conv5.weight', 'layer4.6.conv2.weight', 'layer1.0.conv1.weight'] + ([] if 'A' in tag else ['layer2.0.downsample.0.weight'])
        gn = {k: float(named[k].grad.double().norm()) for k in pick}
        save('fine_' + tag, keys=keys_json(m), logits=y, train_logits=yt, grad_norms=json.dumps(gn),
             **{('g_' + k.replace('.', '_')): thin(named[k].grad) for k in pick[:2]})
    xc, feat, fm, meta, depth = coarse_inputs(170, 1, 16, 12)
    for tp in ('avg', 'max', 'stride', None):
        m = ref_coarse.generate_model('M', n_classes=400, feat_depth=depth, task='loc', dropout=0.0, base_bn_splits=1,
                                      learnedMixing=True, isMixing=True, t_pool=tp)
        m.replace_logits(157)
        spec.fill_module_(m)
        m.eval()
        with torch.no_grad():
            y = m([xc, feat, fm, 0, meta])
        save('coarse_tpool_%s' % tp, keys=keys_json(m), logits=y)


if __name__ == '__main__':
    only = sys.argv[1:]
    cases = [case_keys, case_interp1d, case_gridpool, case_gridunpool, case_gaussian, case_rewight, case_mixing,
             case_subbn, case_bottleneck, case_loss_ap, case_fine, case_coarse, case_multicrop, case_collate, case_options]
    for c in cases:
        if only and c.__name__[5:] not in only:
            continue
        torch.manual_seed(0)
        c()
