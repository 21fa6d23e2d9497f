"""Functional CPU restatement of the reference hot path -- TEST INFRASTRUCTURE ONLY.

All functions take ``sd`` (a mapping key -> fp32 CPU tensor using the reference's
state_dict key names, see oracle/spec.py) and plain tensors.  Stock torch CPU ops
only.  Citations are ``file:line`` under /root/reference.

Parity status: pinned against outputs of the reference itself (imported in the
build container) through tests/golden/*.npz; see tests/test_oracle_golden.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import spec

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# SubBatchNorm3d, x3d_fine.py:13-62
# --------------------------------------------------------------------------------------
def sub_bn(x, sd, p, training, splits=1):
    """Split batch-norm without affine, then one shared affine (x3d_fine.py:51-62).

    Training mode updates ``sd[p+'.split_bn.running_*']`` in place like the module does.
    """
    if training:
        n, c, t, h, w = x.shape
        xv = x.reshape(n // splits, c * splits, t, h, w)
        xv = F.batch_norm(xv, sd[p + '.split_bn.running_mean'], sd[p + '.split_bn.running_var'],
                          None, None, True, BN_MOMENTUM, BN_EPS)
        if (p + '.split_bn.num_batches_tracked') in sd:
            sd[p + '.split_bn.num_batches_tracked'] += 1
        x = xv.reshape(n, c, t, h, w)
    else:
        x = F.batch_norm(x, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'],
                         None, None, False, BN_MOMENTUM, BN_EPS)
    if (p + '.weight') in sd:
        x = x * sd[p + '.weight'].view(-1, 1, 1, 1)
        x = x + sd[p + '.bias'].view(-1, 1, 1, 1)
    return x


def aggregate_bn_stats(sd, splits=1):
    """``aggregate_sub_bn_stats`` (x3d_fine.py:30-49, :321-328) over every SubBN in sd."""
    count = 0
    for k in list(sd.keys()):
        if not k.endswith('.split_bn.running_mean'):
            continue
        p = k[:-len('.split_bn.running_mean')]
        means = sd[k].view(splits, -1)
        vs = sd[p + '.split_bn.running_var'].view(splits, -1)
        mean = means.sum(0) / splits
        var = vs.sum(0) / splits + ((means - mean) ** 2).sum(0) / splits
        sd[p + '.bn.running_mean'] = mean.detach().clone()
        sd[p + '.bn.running_var'] = var.detach().clone()
        count += 1
    return count


def swish(x):
    """x * sigmoid(x); SwishEfficient.backward (x3d_fine.py:82-86) is its exact derivative."""
    return x * torch.sigmoid(x)


# --------------------------------------------------------------------------------------
# Bottleneck, x3d_fine.py:146-175 (= x3d_coarse.py:143-172)
# --------------------------------------------------------------------------------------
def bottleneck(x, sd, p, stride, index, training, splits):
    cm = sd[p + '.conv2.weight'].shape[0]
    out = F.conv3d(x, sd[p + '.conv1.weight'])
    out = F.relu(sub_bn(out, sd, p + '.bn1', training, splits))
    out = F.conv3d(out, sd[p + '.conv2.weight'], stride=(1, stride, stride), padding=1, groups=cm)
    out = sub_bn(out, sd, p + '.bn2', training, splits)
    if index % 2 == 0:  # squeeze-excite over all of (T,H,W), x3d_fine.py:157-163
        se = out.mean(dim=(2, 3, 4), keepdim=True)
        se = F.relu(F.conv3d(se, sd[p + '.fc1.weight'], sd[p + '.fc1.bias']))
        se = torch.sigmoid(F.conv3d(se, sd[p + '.fc2.weight'], sd[p + '.fc2.bias']))
        out = out * se
    out = swish(out)
    out = F.conv3d(out, sd[p + '.conv3.weight'])
    out = sub_bn(out, sd, p + '.bn3', training, splits)
    if (p + '.downsample.0.weight') in sd:
        res = F.conv3d(x, sd[p + '.downsample.0.weight'], stride=(1, stride, stride))
        res = sub_bn(res, sd, p + '.downsample.1', training, splits)
    else:
        res = x
    return F.relu(out + res)


def stem(x, sd, training, splits):
    """conv1_s -> conv1_t -> bn1 -> relu (x3d_fine.py:334-337)."""
    c = sd['conv1_t.weight'].shape[0]
    x = F.conv3d(x, sd['conv1_s.weight'], stride=(1, 2, 2), padding=(0, 1, 1))
    x = F.conv3d(x, sd['conv1_t.weight'], padding=(2, 0, 0), groups=c)
    return F.relu(sub_bn(x, sd, 'bn1', training, splits))


def stage(x, sd, li, version, training, splits):
    for bi in range(spec.BLOCKS[version][li - 1]):
        x = bottleneck(x, sd, 'layer%d.%d' % (li, bi), 2 if bi == 0 else 1, bi, training, splits)
    return x


def head(x, sd, training, splits, dropout_p=0.0):
    """conv5..fc2 for task 'loc' (x3d_fine.py:356-380). ``dropout_p`` > 0 only makes sense
    for timing; parity tests run with dropout disabled (RNG streams differ by device)."""
    x = F.relu(sub_bn(F.conv3d(x, sd['conv5.weight']), sd, 'bn5', training, splits))
    x = x.mean(dim=(3, 4), keepdim=True)
    x = F.relu(F.conv3d(x, sd['fc1.weight']))
    x = x.squeeze(4).squeeze(3).permute(0, 2, 1)
    if training and dropout_p > 0:
        x = F.dropout(x, dropout_p, True)
    return F.linear(x, sd['fc2.weight'], sd['fc2.bias']).permute(0, 2, 1)


def x3d_fine_forward(sd, x, version='M', training=False, splits=1, global_tower=False,
                     dropout_p=0.0):
    """x3d_fine.ResNet.forward, task='loc' (x3d_fine.py:331-382)."""
    x = stem(x, sd, training, splits)
    feats = {}
    for li in range(1, 5):
        x = stage(x, sd, li, version, training, splits)
        if global_tower:
            feats['layer%d' % li] = F.adaptive_avg_pool3d(x, (None, 7, 7))
    if global_tower:
        y = F.relu(sub_bn(F.conv3d(x, sd['conv5.weight']), sd, 'bn5', training, splits))
        feats['conv5'] = F.adaptive_avg_pool3d(y, (None, 7, 7))
        return feats
    return head(x, sd, training, splits, dropout_p)


# --------------------------------------------------------------------------------------
# Interp1d, interp1d.py:8-147
# --------------------------------------------------------------------------------------
def interp1d(x, y, xnew):
    """Batched linear interpolation; returns (ynew, ind) with ``ind`` the int64 left-knot
    index (searchsorted-left minus one, clamped to [0, N-2]; interp1d.py:100-110)."""
    eps = torch.finfo(y.dtype).eps
    X = x[None] if x.dim() == 1 else x
    Y = y[None] if y.dim() == 1 else y
    Q = xnew[None] if xnew.dim() == 1 else xnew
    assert X.shape[1] == Y.shape[1]
    stacked = X.shape[0] == 1 and Y.shape[0] == 1 and Q.shape[0] > 1
    qshape = Q.shape
    if stacked:  # interp1d.py:63-71
        Q = Q.contiguous().view(1, -1)
    if Q.shape[0] == 1:
        Q = Q.expand(X.shape[0], -1)
    ind = torch.searchsorted(X.contiguous(), Q.contiguous()) - 1
    ind = ind.clamp(0, X.shape[1] - 2)

    def sel(v):
        if v.shape[0] == 1:
            return v.contiguous().view(-1)[ind]
        return torch.gather(v, 1, ind)

    slopes = (Y[:, 1:] - Y[:, :-1]) / (eps + (X[:, 1:] - X[:, :-1]))  # interp1d.py:133-137
    ynew = sel(Y) + sel(slopes) * (Q - sel(X))                        # interp1d.py:140-141
    if stacked:
        ynew = ynew.view(qshape)
    return ynew, ind


# --------------------------------------------------------------------------------------
# Grid Pool / Grid Unpool, x3d_coarse.py:355-451
# --------------------------------------------------------------------------------------
def grid_cdf(g):
    """saliency logits (B,K-1) -> CDF knots (B,K) (x3d_coarse.py:384-392)."""
    p = 1. - torch.sigmoid(g * 5e-1)
    p = p / (torch.sum(p, dim=1, keepdim=True) + 1e-16)
    c = torch.cumsum(p, dim=1)
    out = torch.zeros(c.shape[0], c.shape[1] + 1, dtype=torch.float32)
    out[:, 1:] = c
    return out


def _norm_axis(n):
    a = torch.arange(n).to(torch.float32) / (n - 1)
    return (a - 0.5) * 2


def _temporal_grid(coord_t, gh, gw):
    """meshgrid+stack of x3d_coarse.py:400-401 for a (B,K) temporal coordinate."""
    b, k = coord_t.shape
    g = torch.meshgrid([coord_t.reshape(-1), gh, gw], indexing='ij')
    return torch.stack((g[2], g[1], g[0]), dim=-1).view(b, k, gh.shape[0], gw.shape[0], 3)


def grid_sample_time_index(cdf, T):
    """The integer frame index / fractional weight ATen's grid_sampler_3d derives from a
    CDF knot with align_corners=True (GridSampler.h ``grid_sampler_unnormalize``):
    i_t = ((coord + 1) / 2) * (T - 1), i0 = floor(i_t), w1 = i_t - i0."""
    coord = (cdf - 0.5) * 2
    it = ((coord + 1) / 2) * (T - 1)
    i0 = torch.floor(it)
    return i0.to(torch.int32), it - i0


def grid_pool_saliency(x, sd, p, training):
    """conv1/bn1/relu/conv2/bn2/relu/conv3/avg (x3d_coarse.py:379-383) -> (B, T/4)."""
    g = F.conv3d(x, sd[p + '.conv1.weight'], sd[p + '.conv1.bias'], stride=(2, 2, 2), padding=1)
    g = F.relu(sub_bn(g, sd, p + '.bn1', training, 1))
    g = F.conv3d(g, sd[p + '.conv2.weight'], sd[p + '.conv2.bias'], stride=(2, 2, 2), padding=1)
    g = F.relu(sub_bn(g, sd, p + '.bn2', training, 1))
    g = F.conv3d(g, sd[p + '.conv3.weight'], sd[p + '.conv3.bias'], stride=(1, 2, 2), padding=(0, 1, 1))
    return g.mean(dim=(3, 4)).squeeze(1)


def grid_pool_resample(x, cdf):
    """x (B,C,T,H,W), cdf (B,K) -> (B,C,K,H,W) via 5-D grid_sample (x3d_coarse.py:394-403)."""
    h, w = x.shape[3], x.shape[4]
    grid = _temporal_grid((cdf - 0.5) * 2, _norm_axis(h), _norm_axis(w))
    return F.grid_sample(x, grid, align_corners=True)


def grid_pool(x, sd, p, training):
    """GridPoolLayer.forward (x3d_coarse.py:373-416) -> (x_pooled, cdf)."""
    cdf = grid_cdf(grid_pool_saliency(x, sd, p, training))
    return grid_pool_resample(x, cdf), cdf


def grid_unpool(x, cdf, is_logit, ratio=4):
    """GridUnpool (x3d_coarse.py:419-451); also returns the Interp1d indices."""
    b = x.shape[0]
    k = cdf.shape[1]
    mid = torch.arange(k).to(torch.float32)
    mid = (mid / (k - 1.)).view(1, -1).repeat(b, 1)
    inv, ind = interp1d(cdf, mid, mid)
    if is_logit:
        x5 = x.unsqueeze(3).unsqueeze(4)
        gh = gw = torch.zeros(1, dtype=torch.float32)
    else:
        x5 = x
        gh, gw = _norm_axis(x.shape[3]), _norm_axis(x.shape[4])
    out = F.grid_sample(x5, _temporal_grid((inv - 0.5) * 2, gh, gw), align_corners=True)
    if is_logit:
        out = out.squeeze(4).squeeze(3)
    else:
        t, h, w = x.shape[2:]
        out = F.interpolate(out, (t * ratio, h, w), mode='trilinear', align_corners=True)
    return out, inv, ind


# --------------------------------------------------------------------------------------
# Multi-stage fusion, x3d_coarse.py:175-351
# --------------------------------------------------------------------------------------
def gaussian(meta, mask, gx, tx, ratio=1):
    """Gaussian.forward (x3d_coarse.py:256-286): GX (B, T', K)."""
    st, step = meta[:, 0], meta[:, 3]
    b, b2, len_f = meta.shape[0], gx.shape[0], mask.shape[1]
    st = st.to(torch.float32)
    if b2 != b:  # multi-crop testing, :264-266
        off = step.view(-1, 1) * torch.arange(0, b2 // b).to(torch.float32).view(1, -1).repeat(b, 1)
        st = (st.view(-1, 1).repeat(1, b2 // b) + off).view(-1, 1)
    if tx is not None:
        len_x = gx.shape[1]
        tl = (gx * tx).unsqueeze(1)
    else:
        len_x = gx.shape[2]
        tl = torch.arange(0, len_x).to(torch.float32).view(1, 1, -1).repeat(b2, 1, 1)
    mu = (tl + st.view(b2, 1, 1)) / ratio
    t = torch.arange(0, len_f).to(torch.float32).view(1, -1, 1).repeat(b2, 1, 1)
    std = (1 / 8 * torch.sum(mask, dim=1)).view(-1, 1).repeat(1, b2 // b).view(-1, 1)
    t = t - mu
    f = t ** 2 / (2 * (std ** 2).view(b2, 1, 1).repeat(1, len_f, len_x) + 1e-16)
    f = torch.exp(-f)
    f = f / (torch.max(f, dim=1)[0].view(b2, 1, len_x) + 1e-16)
    return f.view(b2, len_f, len_x)


def _conv1d(x, sd, p):
    return F.conv1d(x, sd[p + '.weight'], sd[p + '.bias'])


def rewight(sd, p, x, lx_shape, mask, GX, is_mixing, height, pool=False):
    """RewightLayer.forward (x3d_coarse.py:199-247) without the 6-D materialisation: the
    (t)-sum of x*at*GX*mask / (sum at*GX*mask + 1e-6) is taken by einsum; same arithmetic,
    different summation order.  Dropout (pool=True, :232-233) is left out: parity runs
    disable it."""
    b, c, t, h, w = x.shape
    b2, tl = lx_shape[0], lx_shape[2]
    hl = wl = height
    if mask.shape[1] != t:
        mask = F.adaptive_max_pool1d(mask.unsqueeze(1), t).squeeze(1)
        GX = F.adaptive_avg_pool2d(GX.unsqueeze(1), (t, None)).squeeze(1)
    if b != b2:
        x = x.unsqueeze(1).repeat(1, b2 // b, 1, 1, 1, 1).view(b2, c, t, h, w)
        mask = mask.unsqueeze(1).repeat(1, b2 // b, 1).view(b2, t)
    if h != hl:
        x = F.adaptive_max_pool2d(x.reshape(b2, c * t, h, w), (hl, wl)).view(b2, c, t, hl, wl)
    at = F.relu(_conv1d(x.reshape(b2, c, -1), sd, p + '.at1'))
    at = torch.sigmoid(_conv1d(at, sd, p + '.at2')).view(b2, t, hl, wl)
    wgt = at.unsqueeze(2) * GX.view(b2, t, tl, 1, 1) * mask.view(b2, t, 1, 1, 1)   # B T Tl H W
    den = wgt.sum(dim=1) + 1e-6                                                    # B Tl H W
    num = torch.einsum('bcthw,btkhw->bckhw', x, wgt)
    z = num / den.unsqueeze(1)
    if pool:
        z = F.adaptive_avg_pool3d(z, (None, 1, 1))
    bb, cc, tt, hh, ww = z.shape
    x1 = _conv1d(F.relu(_conv1d(z.reshape(bb, cc, -1), sd, p + '.fc1')), sd, p + '.fc2')
    x1 = x1.view(bb, -1, tt, hh, ww)
    x2 = _conv1d(F.relu(_conv1d(z.reshape(bb, cc, -1), sd, p + '.fc3')), sd, p + '.fc4')
    x2 = x2.view(bb, -1, tt, hh, ww)
    if not is_mixing:
        x2 = torch.sigmoid(x2)
    return x1, x2


def mixing(sd, p, x_shape, bias, scale):
    """MixingLayer.forward, learned path (x3d_coarse.py:307-336) -> (c, m) shaped like x."""
    b, c, t, h, w = x_shape

    def gather(lst):
        out = []
        for v in lst:
            _, cf, _, hf, wf = v.shape
            if hf != h:
                v = F.adaptive_max_pool2d(v.reshape(b, cf * t, hf, wf), (h, w)).view(b, cf, t, h, w)
            out.append(v)
        return torch.cat(out, dim=1)

    cs, ms = gather(bias), gather(scale)
    cs = _conv1d(cs.reshape(b, -1, t * h * w), sd, p + '.conv_at').view(b, c, t, h, w)
    ms = torch.sigmoid(_conv1d(ms.reshape(b, -1, t * h * w), sd, p + '.conv_at2')).view(b, c, t, h, w)
    return cs, ms


def x3d_coarse_forward(sd, inp, version='M', training=False, splits=1, is_mixing=True,
                       return_aux=False):
    """x3d_coarse.ResNet.forward with t_pool='grid', learnedMixing=True, task='loc'
    (x3d_coarse.py:628-727)."""
    x, feat, feat_masks, _, meta = inp
    tl = x.shape[2]
    x = stem(x, sd, training, splits)
    x = stage(x, sd, 1, version, training, splits)
    x, cdf = grid_pool(x, sd, 'pool_1', training)
    GX = gaussian(meta, feat_masks, cdf, tl)
    fkeys = ('layer1', 'layer2', 'layer3', 'layer4')
    if is_mixing:
        rb, rs = [], []
        for i, k, hgt in zip(range(2, 6), fkeys, spec.FUSION_HEIGHTS):
            b_, s_ = rewight(sd, 'rw%d' % i, feat[k], x.shape, feat_masks, GX, True, hgt)
            rb.append(b_)
            rs.append(s_)
        for li in range(2, 6):
            c_, m_ = mixing(sd, 'mix%d' % li, x.shape, rb, rs)
            x = x * m_ + c_
            if li < 5:
                x = stage(x, sd, li, version, training, splits)
    else:
        for li, k, hgt in zip(range(2, 6), fkeys, spec.FUSION_HEIGHTS):
            b_, s_ = rewight(sd, 'rw%d' % li, feat[k], x.shape, feat_masks, GX, False, hgt)
            x = x * s_ + b_
            if li < 5:
                x = stage(x, sd, li, version, training, splits)
    x = head(x, sd, training, splits)                       # (B, n_cls, K)
    x5 = x.unsqueeze(3).unsqueeze(4)
    b6, s6 = rewight(sd, 'rw6', feat['conv5'], x5.shape, feat_masks, GX, False, 7, pool=True)
    x = (x5 * s6 + b6).squeeze(4).squeeze(3)
    pooled_logits = x
    x, inv, ind = grid_unpool(x, cdf, True)
    x = F.interpolate(x, (x.shape[2] - 1) * 4, mode='linear', align_corners=True)
    if return_aux:
        return x, {'cdf': cdf, 'GX': GX, 'inv': inv, 'ind': ind, 'pooled_logits': pooled_logits}
    return x


# --------------------------------------------------------------------------------------
# loss + AP (harness pieces needed for parity statements)
# --------------------------------------------------------------------------------------
def detection_loss(logits, labels, masks, align_corners, crops=1):
    """cls/loc loss of train_fine.py:199-213 (align_corners=True) and
    train_coarse_fineFEAT.py:226-240 (align_corners=False).  crops = n > 1: the validation branch
    (train_fine.py:204-207, train_coarse_fineFEAT.py:231-235): logits (b*n, C, T), max over the n crops."""
    tl = labels.shape[2]
    lg = F.interpolate(logits, tl, mode='linear', align_corners=align_corners)
    if crops > 1:
        probs = torch.max(torch.sigmoid(lg.view(labels.shape[0], crops, -1, tl)), dim=1)[0] * masks.unsqueeze(1)
    else:
        probs = torch.sigmoid(lg) * masks.unsqueeze(1)
    cls = F.binary_cross_entropy(torch.max(probs, dim=2)[0], torch.max(labels, dim=2)[0])
    loc = F.binary_cross_entropy(probs, labels, reduction='sum') / (torch.sum(masks) * labels.shape[1])
    return cls, loc, probs


def average_precision(scores, targets):
    """APMeter.value (apmeter.py:98-136) for unweighted samples: numpy, (N,K) -> (K,)."""
    scores = np.asarray(scores, dtype=np.float32)
    targets = np.asarray(targets)
    n, k = scores.shape
    ap = np.zeros(k, dtype=np.float32)
    rg = np.arange(1, n + 1, dtype=np.float32)
    for j in range(k):
        order = torch.sort(torch.from_numpy(scores[:, j].copy()), 0, True)[1].numpy()
        truth = targets[order, j].astype(np.float32)
        prec = np.cumsum(truth, dtype=np.float32) / rg
        ap[j] = prec[truth > 0].sum() / max(truth.sum(), 1)
    return ap
