"""Host-side batch builders against the behaviour of the reference's mt_collate_fn
(charades_fine.py:201-224, charades_coarse_fineFEAT.py:208-252): right zero padding to the batch maximum, per-sample
label masks, fine features and their mask capped at 128 frames."""
import numpy as np
import torch

import collate


def _fine_sample(t, tl, seed):
    r = np.random.RandomState(seed)
    return r.randn(1, 3, t, 8, 8).astype(np.float32), (r.rand(157, tl) < 0.1).astype(np.float32), 'vid%d' % seed


def test_fine_collate_pads_and_masks():
    batch = [_fine_sample(5, 50, 0), _fine_sample(8, 80, 1), _fine_sample(3, 30, 2)]
    clips, label, mask, vids = collate.fine_collate(batch)
    assert clips.shape == (3, 1, 3, 8, 8, 8) and label.shape == (3, 157, 80) and mask.shape == (3, 80)
    assert clips.dtype == label.dtype == mask.dtype == torch.float32
    for i, (c, lb, v) in enumerate(batch):
        t, tl = c.shape[2], lb.shape[1]
        assert torch.equal(clips[i, :, :, :t], torch.from_numpy(c)) and float(clips[i, :, :, t:].abs().sum()) == 0.0
        assert torch.equal(label[i, :, :tl], torch.from_numpy(lb)) and float(label[i, :, tl:].abs().sum()) == 0.0
        assert mask[i].tolist() == [1.0] * tl + [0.0] * (80 - tl)
    assert vids == ['vid0', 'vid1', 'vid2']


def _coarse_sample(t, tl, tf, seed):
    r = np.random.RandomState(seed)
    depth = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}
    feat = {k: np.abs(r.randn(c, tf, 7, 7)).astype(np.float32) for k, c in depth.items()}
    meta = np.array([seed, t, tf, 1], dtype=np.int64)
    return (r.randn(1, 3, t, 8, 8).astype(np.float32), (r.rand(157, tl) < 0.1).astype(np.float32), feat, meta,
            'vid%d' % seed, 10.0 + seed)


def test_coarse_collate_caps_fine_features():
    batch = [_coarse_sample(4, 40, 150, 0), _coarse_sample(6, 60, 70, 1)]
    clips, label, mask, feat, fmask, meta, vids, dur = collate.coarse_collate(batch)
    assert clips.shape == (2, 1, 3, 6, 8, 8) and label.shape == (2, 157, 60) and mask.shape == (2, 60)
    assert fmask.shape == (2, 128) and sorted(feat) == sorted(batch[0][2])
    for k, f in feat.items():
        assert f.shape == (2, batch[0][2][k].shape[0], 128, 7, 7)
        assert torch.equal(f[0], torch.from_numpy(batch[0][2][k][:, :128]))             # truncated at the cap
        assert torch.equal(f[1, :, :70], torch.from_numpy(batch[1][2][k])) and float(f[1, :, 70:].abs().sum()) == 0.0
    assert fmask[0].tolist() == [1.0] * 128 and fmask[1].tolist() == [1.0] * 70 + [0.0] * 58
    assert mask[0].tolist() == [1.0] * 40 + [0.0] * 20
    assert meta.dtype == torch.int64 and meta.tolist() == [[0, 4, 150, 1], [1, 6, 70, 1]]
    assert vids == ['vid0', 'vid1'] and dur.tolist() == [10.0, 11.0]


def test_coarse_collate_below_cap_keeps_batch_maximum():
    batch = [_coarse_sample(4, 40, 33, 0), _coarse_sample(4, 40, 21, 1)]
    _, _, _, feat, fmask, _, _, _ = collate.coarse_collate(batch)
    assert fmask.shape == (2, 33) and feat['conv5'].shape == (2, 432, 33, 7, 7)
    assert fmask[1].tolist() == [1.0] * 21 + [0.0] * 12
