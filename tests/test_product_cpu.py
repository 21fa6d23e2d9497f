"""Host-side product code pinned to outputs of the reference (CPU, no HIP calls): the loss of both training scripts, the
two batch builders, the per-GPU hand-off structures.  Fixtures: tests/golden/make_golden.py (loss_ap, loss_multicrop,
collate_fine, collate_coarse)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, t, maxdiff


@pytest.mark.parametrize('ac', [1, 0])
def test_product_detection_loss_matches_reference(ac):
    """train_fine.detection_loss (align_corners=True, train_fine.py:199-213) and train_coarse_fineFEAT.detection_loss
    (no align_corners, train_coarse_fineFEAT.py:226-240) against the losses the reference's own expressions gave"""
    import train_fine
    import train_coarse_fineFEAT as tc
    z = load_golden('loss_ap')
    lg, labels, masks = t(z['logits']), t(z['labels']), t(z['masks'])
    if ac:
        cls, loc, _ = train_fine.detection_loss(lg, labels, masks, True)
    else:
        cls, loc, _ = tc.detection_loss(lg, labels, masks)
    assert abs(float(cls) - float(z['cls_%d' % ac])) <= 1e-6
    assert abs(float(loc) - float(z['loc_%d' % ac])) <= 1e-6


@pytest.mark.parametrize('ac', [1, 0])
def test_product_detection_loss_multicrop(ac):
    """validation branch: n crops per video, max over crops (train_fine.py:204-207)"""
    import train_fine
    import train_coarse_fineFEAT as tc
    z = load_golden('loss_multicrop')
    lg, labels, masks, n = t(z['logits']), t(z['labels']), t(z['masks']), int(z['crops'])
    if ac:
        cls, loc, probs = train_fine.detection_loss(lg, labels, masks, True, crops=n, local_norm=True)
    else:
        cls, loc, probs = tc.detection_loss(lg, labels, masks, crops=n, local_norm=True)
    assert probs.shape == labels.shape
    assert abs(float(cls) - float(z['cls_%d' % ac])) <= 1e-6 and abs(float(loc) - float(z['loc_%d' % ac])) <= 1e-6
    assert maxdiff(probs[:, ::13], z['probs_%d' % ac]) <= 1e-6


def test_fine_collate_equals_reference_mt_collate_fn():
    import collate
    z = load_golden('collate_fine')
    batch = [(z['in%d_clips' % i], z['in%d_label' % i], 'vid%d' % i) for i in range(3)]
    clips, label, mask, vids = collate.fine_collate(batch)
    for mine, key in ((clips, 'clips'), (label, 'label'), (mask, 'mask')):
        assert mine.dtype == torch.float32 and tuple(mine.shape) == z[key].shape and torch.equal(mine, t(z[key])), key
    assert list(vids) == json.loads(str(z['vids']))


def test_coarse_collate_equals_reference_mt_collate_fn():
    import collate
    z = load_golden('collate_coarse')
    keys = ('layer1', 'conv5')
    batch = [(z['in%d_clips' % i], z['in%d_label' % i], {k: z['in%d_feat_%s' % (i, k)] for k in keys}, z['in%d_meta' % i],
              'vid%d' % i, 10.5 + i) for i in range(3)]
    clips, label, mask, feat, fmask, meta, vids, dur = collate.coarse_collate(batch)
    for mine, key in ((clips, 'clips'), (label, 'label'), (mask, 'mask'), (fmask, 'fmask'), (meta, 'meta'), (dur, 'dur')):
        ref = t(z[key])
        assert mine.dtype == ref.dtype and torch.equal(mine, ref), key
    for k in keys:
        assert torch.equal(feat[k], t(z['feat_' + k])), k
    assert fmask.shape[1] == 128 and float(fmask[0].sum()) == 100.0       # capped at 128, sample 0 is 100 frames long
    assert list(vids) == json.loads(str(z['vids']))
