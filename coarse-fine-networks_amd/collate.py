"""Batch builders of the two training scripts: ragged per-video samples -> the padded, masked batch
structure the models and losses consume.

* ``fine_collate``  = ``mt_collate_fn`` of charades_fine.py:201-224
  sample  = (clips (n,3,T,H,W), label (157,TL), vid)
  batch   = [clips (B,n,3,Tmax,H,W), label (B,157,TLmax), mask (B,TLmax), [vid...]]
* ``coarse_collate`` = ``mt_collate_fn`` of charades_coarse_fineFEAT.py:208-252
  sample  = (clips, label, feat{k: (C_k,T',7,7)}, meta (4,), vid, dur)
  batch   = [clips, label, mask, feat{k: (B,C_k,T'max<=cap,7,7)}, feat_mask (B,T'max), meta (B,4), [vid...], dur (B,)]

Zero padding on the right along time, ``mask`` = 1 over each sample's own label length, fine features and their mask
truncated to ``cap`` = 128 frames (the Gaussian-alignment tables of the fusion layers are sized for that).  Inputs may
be numpy arrays or tensors; outputs are fp32 tensors (pin them and copy with non_blocking=True in the loader).
"""
import numpy as np
import torch


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))


def _pad_time(items, dim, length):
    """stack `items` after zero-padding dimension `dim` on the right to `length`"""
    first = _t(items[0])
    shape = list(first.shape)
    shape[dim] = length
    out = torch.zeros([len(items)] + shape, dtype=torch.float32)
    for i, it in enumerate(items):
        it = _t(it).to(torch.float32)
        n = min(it.shape[dim], length)
        idx = [i] + [slice(None)] * it.dim()
        idx[dim + 1] = slice(0, n)
        out[tuple(idx)] = it.narrow(dim, 0, n)
    return out


def _label_mask(labels, tl):
    mask = torch.zeros(len(labels), tl, dtype=torch.float32)
    for i, lb in enumerate(labels):
        mask[i, :_t(lb).shape[1]] = 1.0
    return mask


def fine_collate(batch):
    clips = [b[0] for b in batch]
    labels = [b[1] for b in batch]
    t_max = max(_t(c).shape[2] for c in clips)
    tl_max = max(_t(lb).shape[1] for lb in labels)
    return [_pad_time(clips, 2, t_max), _pad_time(labels, 1, tl_max), _label_mask(labels, tl_max), [b[2] for b in batch]]


def coarse_collate(batch, cap=128):
    clips = [b[0] for b in batch]
    labels = [b[1] for b in batch]
    feats = [b[2] for b in batch]
    keys = list(feats[0].keys())
    t_max = max(_t(c).shape[2] for c in clips)
    tl_max = max(_t(lb).shape[1] for lb in labels)
    tf_max = min(max(_t(f[keys[0]]).shape[1] for f in feats), cap)
    feat = {k: _pad_time([f[k] for f in feats], 1, tf_max) for k in keys}
    feat_mask = torch.zeros(len(batch), tf_max, dtype=torch.float32)
    for i, f in enumerate(feats):
        feat_mask[i, :min(cap, _t(f[keys[0]]).shape[1])] = 1.0
    meta = torch.stack([_t(b[3]) for b in batch])
    dur = torch.as_tensor([float(b[5]) for b in batch], dtype=torch.float64)
    return [_pad_time(clips, 2, t_max), _pad_time(labels, 1, tl_max), _label_mask(labels, tl_max), feat, feat_mask, meta,
            [b[4] for b in batch], dur]
