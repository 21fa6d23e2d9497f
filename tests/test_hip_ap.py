"""AP / mAP parity (north_star: "per-clip logits/APs match the reference CPU path ... mAP within +-0.1 of
reference on identical inputs").

The HIP models' per-frame scores and the CPU oracle's go through the SAME validation arithmetic the reference scripts use
(train_fine.py:197-218: linear resize to the label length with align_corners=True, sigmoid, mask, per-video `valid_t` slice;
train_coarse_fineFEAT.py:226-266: default align_corners, then the 25-frame Charades_v1_localize subsampling
`p1[:, 1::int(valid_t/25)][:, :25]`) and through `apmeter.APMeter` (pinned to the reference's AP vector in
tests/test_product_cpu.py) with the same Bernoulli labels.  AP is rank based: a score difference matters only where it swaps a
positive with a negative, so the bound on the AP vector is stated per class and on the mean."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, t

pytestmark = pytest.mark.gpu
DEV = 'cuda'
NCLS = 157


def _labels(seed, B, tl, p=0.08):
    g = torch.Generator().manual_seed(seed)
    labels = (torch.rand(B, NCLS, tl, generator=g) < p).float()
    valid = torch.randint(tl // 2, tl + 1, (B,), generator=g)
    masks = torch.zeros(B, tl)
    for b in range(B):
        masks[b, :int(valid[b])] = 1
        labels[b, :, int(valid[b]):] = 0
    return labels, masks, valid


def _ap_fine(probs, labels, valid):
    """train_fine.py:214-218 (val branch, one crop)"""
    import apmeter
    apm = apmeter.APMeter()
    for b in range(labels.shape[0]):
        v = int(valid[b])
        apm.add(probs[b][:, :v].transpose(0, 1).detach().cpu().numpy(), labels[b][:, :v].transpose(0, 1).cpu().numpy())
    return apm.value().double()


def _ap_coarse(probs, labels, valid):
    """train_coarse_fineFEAT.py:249-266: 25 equally spaced frames per video"""
    import apmeter
    apm = apmeter.APMeter()
    for b in range(labels.shape[0]):
        v = int(valid[b])
        p1, l1 = probs[b][:, :v], labels[b][:, :v]
        sc = v / 25.
        p1, l1 = p1[:, 1::int(sc)][:, :25], l1[:, 1::int(sc)][:, :25]
        apm.add(p1.transpose(0, 1).detach().cpu().numpy(), l1.transpose(0, 1).cpu().numpy())
    return apm.value().double()


def _check(ap_hip, ap_ref, what):
    d = (ap_hip - ap_ref).abs()
    dmap = abs(float(ap_hip.mean() - ap_ref.mean())) * 100
    print('%s: mAP hip %.4f %% / oracle %.4f %%, |dmAP| = %.5f pt, per-class max |dAP| = %.2e, classes with |dAP| > 1e-4: %d of %d'
          % (what, float(ap_hip.mean()) * 100, float(ap_ref.mean()) * 100, dmap, float(d.max()), int((d > 1e-4).sum()), NCLS))
    assert float(ap_ref.mean()) > 0.02                      # the labels are not degenerate
    assert dmap <= 0.1, what                                # north_star: mAP within +-0.1 (points)
    assert float(d.max()) <= 1e-3, what                     # per-class AP


def test_fine_ap_vs_oracle():
    """x3d_fine X3D-M, 8 clips x 16 frames x 224^2, eval mode (the reference computes val APs under net.train(False))"""
    import x3d_fine
    import train_fine
    from oracle import spec, x3d_ref
    B, T, tl = 8, 16, 64
    net = x3d_fine.generate_model('M', n_classes=NCLS, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(net)
    net.to(DEV).eval()
    x = spec.rand_input(301, (B, 3, T, 224, 224))
    labels, masks, valid = _labels(302, B, tl)
    with torch.no_grad():
        logits = net([x.to(DEV), None])
        _, _, probs = train_fine.detection_loss(logits, labels.to(DEV), masks.to(DEV), align_corners=True, local_norm=True)
        sd = spec.procedural_fill(spec.fine_keys('M', NCLS, 1))
        lo = torch.cat([x3d_ref.x3d_fine_forward(sd, x[b:b + 1], 'M', training=False) for b in range(B)])
        po = torch.sigmoid(F.interpolate(lo, tl, mode='linear', align_corners=True)) * masks.unsqueeze(1)
    assert float((logits.cpu() - lo).abs().max()) <= 1e-3
    _check(_ap_fine(probs.cpu(), labels, valid), _ap_fine(po, labels, valid), 'fine')


def test_coarse_ap_vs_oracle():
    """x3d_coarse (fineFEAT fusion, Grid Pool) X3D-M, 8 clips x 32 frames, fine features T' = 24, eval mode; the 25-frame
    Charades_v1_localize subsampling of the coarse script on both sides"""
    import x3d_coarse
    import train_coarse_fineFEAT as tc
    from oracle import spec, x3d_ref
    B, T, Tf, tl = 8, 32, 24, 128
    depth = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}
    net = x3d_coarse.generate_model('M', n_classes=400, feat_depth=depth, task='loc', dropout=0.5, base_bn_splits=1,
                                    learnedMixing=True, isMixing=True, t_pool='grid')
    net.replace_logits(NCLS)
    spec.fill_module_(net)
    net.to(DEV).eval()
    x = spec.rand_input(311, (B, 3, T, 224, 224))
    feat = {k: spec.rand_input(312 + i, (B, c, Tf, 7, 7), nonneg=True) for i, (k, c) in enumerate(depth.items())}
    fm = torch.ones(B, Tf)
    meta = torch.zeros(B, 4, dtype=torch.int64)
    for b in range(B):
        nf = Tf - (b % 4)
        fm[b, nf:] = 0
        meta[b] = torch.tensor([b % 3, T, nf, 1])
    labels, masks, valid = _labels(320, B, tl)
    with torch.no_grad():
        logits = net([x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), 0, meta.to(DEV)])
        _, _, probs = tc.detection_loss(logits, labels.to(DEV), masks.to(DEV), local_norm=True)
        sd = spec.procedural_fill(spec.coarse_keys('M', NCLS, 1))
        lo = torch.cat([x3d_ref.x3d_coarse_forward(sd, [x[b:b + 1], {k: v[b:b + 1] for k, v in feat.items()}, fm[b:b + 1], 0,
                                                        meta[b:b + 1]], 'M', training=False) for b in range(B)])
        po = torch.sigmoid(F.interpolate(lo, tl, mode='linear')) * masks.unsqueeze(1)
    assert float((logits.cpu() - lo).abs().max()) <= 1e-3
    rows = tc.localize_rows(probs.cpu(), labels, valid, ['vid'], torch.full((B,), 30.0))
    import apmeter
    apm = apmeter.APMeter()
    for _, sc, tg in rows:                                      # the product's own localize path ...
        apm.add(sc, tg)
    ap_prod = apm.value().double()
    ap_hip = _ap_coarse(probs.cpu(), labels, valid)             # ... equals the reference's spelling of it
    assert torch.equal(ap_prod, ap_hip)
    _check(ap_hip, _ap_coarse(po, labels, valid), 'coarse')


@pytest.mark.parametrize('name,kind', [('fine_cfg1', 'fine'), ('coarse_eval', 'coarse')])
def test_ap_from_reference_golden_logits(name, kind):
    """the same comparison against logits captured from the REFERENCE itself (tests/golden/*.npz): the HIP model's scores and the
    reference's give the same AP vector on the same labels"""
    from oracle import spec
    z = load_golden(name)
    ref = t(z['logits']).float()
    if kind == 'fine':
        import x3d_fine
        from conftest import golden_sd
        net = x3d_fine.generate_model('S', n_classes=400, task='loc', base_bn_splits=1)
        net.replace_logits(NCLS)
        net.load_state_dict({k: v.clone() for k, v in golden_sd(z).items()})
        net.to(DEV).eval()
        with torch.no_grad():
            y = net([spec.rand_input(0, (1, 3, 13, 160, 160)).to(DEV), None]).cpu()
    else:
        from test_hip_models import _coarse_inputs, _coarse_model
        x, feat, fm, meta, depth = _coarse_inputs(100, 1, 16, 12)
        m = _coarse_model(depth).eval()
        with torch.no_grad():
            y = m([x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), 0, meta.to(DEV)]).cpu()
    assert y.shape == ref.shape and float((y - ref).abs().max()) <= 1e-3
    tl = 4 * y.shape[2]
    labels, masks, valid = _labels(330, 1, tl, p=0.2)
    ac = kind == 'fine'
    ph = torch.sigmoid(F.interpolate(y, tl, mode='linear', align_corners=ac)) * masks.unsqueeze(1)
    pr = torch.sigmoid(F.interpolate(ref, tl, mode='linear', align_corners=ac)) * masks.unsqueeze(1)
    ah, ar = _ap_fine(ph, labels, valid), _ap_fine(pr, labels, valid)
    d = (ah - ar).abs()
    print('%s: mAP hip %.4f / reference %.4f, max |dAP| %.2e' % (name, float(ah.mean()), float(ar.mean()), float(d.max())))
    assert abs(float(ah.mean() - ar.mean())) * 100 <= 0.1
    assert float(d.max()) <= 1e-3
