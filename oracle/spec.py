"""Architecture tables, state_dict key/shape enumeration and procedural weights.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference keeps its weights in ``nn.Module`` attributes whose dotted names
are the checkpoint contract ("checkpoint / resume").  The
oracle is functional, so this file restates *which* keys exist and their shapes:

  * width/depth tables ........ x3d_fine.py:388-400 (= x3d_coarse.py:733-745)
  * SE width rounding ......... x3d_fine.py:132-143
  * SubBatchNorm3d buffers .... x3d_fine.py:13-28
  * Bottleneck members ........ x3d_fine.py:108-130
  * fine ResNet members ....... x3d_fine.py:181-258
  * coarse ResNet members ..... x3d_coarse.py:457-555, GridPoolLayer :355-369,
                                RewightLayer :175-196, MixingLayer :289-304

``procedural_fill`` is the deterministic by-key-name weight generator of
SURVEY.md Appendix A, so fixtures store only seeds and outputs.
"""
from collections import OrderedDict
import zlib

import numpy as np
import torch

PLANES = {
    'S': [(54, 24), (108, 48), (216, 96), (432, 192)],
    'M': [(54, 24), (108, 48), (216, 96), (432, 192)],
    'XL': [(72, 32), (162, 72), (306, 136), (630, 280)],
}
BLOCKS = {'S': [3, 5, 11, 7], 'M': [3, 5, 11, 7], 'XL': [5, 10, 25, 15]}
FUSION_HEIGHTS = (56, 28, 14, 7)  # x3d_coarse.py:535-538, hard coded


def se_width(width, multiplier=0.0625, min_width=8, divisor=8):
    """Squeeze width of the SE branch (x3d_fine.py:132-143)."""
    w = width * multiplier
    out = max(min_width, int(w + divisor / 2) // divisor * divisor)
    if out < 0.9 * w:
        out += divisor
    return int(out)


def _subbn(keys, p, c, splits):
    keys[p + '.weight'] = (c,)
    keys[p + '.bias'] = (c,)
    keys[p + '.bn.running_mean'] = (c,)
    keys[p + '.bn.running_var'] = (c,)
    keys[p + '.bn.num_batches_tracked'] = ()
    keys[p + '.split_bn.running_mean'] = (c * splits,)
    keys[p + '.split_bn.running_var'] = (c * splits,)
    keys[p + '.split_bn.num_batches_tracked'] = ()


def _trunk(keys, version, splits, n_in=3):
    planes = PLANES[version]
    stem = planes[0][1]
    keys['conv1_s.weight'] = (stem, n_in, 1, 3, 3)
    keys['conv1_t.weight'] = (stem, 1, 5, 1, 1)
    _subbn(keys, 'bn1', stem, splits)
    cin = stem
    for li, ((cm, co), nb) in enumerate(zip(planes, BLOCKS[version]), start=1):
        for bi in range(nb):
            p = 'layer%d.%d' % (li, bi)
            keys[p + '.conv1.weight'] = (cm, cin, 1, 1, 1)
            _subbn(keys, p + '.bn1', cm, splits)
            keys[p + '.conv2.weight'] = (cm, 1, 3, 3, 3)
            _subbn(keys, p + '.bn2', cm, splits)
            keys[p + '.conv3.weight'] = (co, cm, 1, 1, 1)
            _subbn(keys, p + '.bn3', co, splits)
            if bi % 2 == 0:
                w = se_width(cm)
                keys[p + '.fc1.weight'] = (w, cm, 1, 1, 1)
                keys[p + '.fc1.bias'] = (w,)
                keys[p + '.fc2.weight'] = (cm, w, 1, 1, 1)
                keys[p + '.fc2.bias'] = (cm,)
            if bi == 0:  # stride 2 on every stage => shortcut type 'B' conv
                keys[p + '.downsample.0.weight'] = (co, cin, 1, 1, 1)
                _subbn(keys, p + '.downsample.1', co, splits)
            cin = co
    keys['conv5.weight'] = (planes[3][0], planes[3][1], 1, 1, 1)
    _subbn(keys, 'bn5', planes[3][0], splits)
    keys['fc1.weight'] = (2048, planes[3][0], 1, 1, 1)


def fine_keys(version='M', n_classes=157, base_bn_splits=1):
    """(key -> shape) of ``x3d_fine.generate_model(version).state_dict()``."""
    keys = OrderedDict()
    _trunk(keys, version, base_bn_splits)
    keys['fc2.weight'] = (n_classes, 2048)
    keys['fc2.bias'] = (n_classes,)
    return keys


def _rewight(keys, p, channels, depth):
    for nm, (o, i) in (('at1', (depth, depth)), ('at2', (1, depth)),
                       ('fc1', (depth, depth)), ('fc2', (channels, depth)),
                       ('fc3', (depth, depth)), ('fc4', (channels, depth))):
        keys['%s.%s.weight' % (p, nm)] = (o, i, 1)
        keys['%s.%s.bias' % (p, nm)] = (o,)


def coarse_keys(version='M', n_classes=157, base_bn_splits=1, feat_depth=None):
    """Keys of ``x3d_coarse.generate_model(version, t_pool='grid',
    learnedMixing=True, isMixing=True)`` after ``replace_logits(n_classes)``."""
    planes = PLANES[version]
    if feat_depth is None:
        feat_depth = {'layer1': planes[0][1], 'layer2': planes[1][1], 'layer3': planes[2][1],
                      'layer4': planes[3][1], 'conv5': planes[3][0]}
    keys = OrderedDict()
    d = planes[0][1]
    keys['pool_1.conv1.weight'] = (d, d, 3, 3, 3)
    keys['pool_1.conv1.bias'] = (d,)
    _subbn(keys, 'pool_1.bn1', d, 1)
    keys['pool_1.conv2.weight'] = (d, d, 3, 3, 3)
    keys['pool_1.conv2.bias'] = (d,)
    _subbn(keys, 'pool_1.bn2', d, 1)
    keys['pool_1.conv3.weight'] = (1, d, 1, 3, 3)
    keys['pool_1.conv3.bias'] = (1,)
    _trunk(keys, version, base_bn_splits)
    for i, lk in zip(range(2, 6), ('layer1', 'layer2', 'layer3', 'layer4')):
        _rewight(keys, 'rw%d' % i, planes[i - 2][1], feat_depth[lk])
    _rewight(keys, 'rw6', n_classes, feat_depth['conv5'])
    mix_in = sum(p[1] for p in planes)  # 24+48+96+192 hard coded at x3d_coarse.py:297
    for i in range(2, 6):
        for nm in ('conv_at', 'conv_at2'):
            keys['mix%d.%s.weight' % (i, nm)] = (planes[i - 2][1], mix_in, 1)
            keys['mix%d.%s.bias' % (i, nm)] = (planes[i - 2][1],)
    keys['fc2.weight'] = (n_classes, 2048)
    keys['fc2.bias'] = (n_classes,)
    return keys


def procedural_value(key, shape):
    """Deterministic fp32 tensor for one state_dict key."""
    if key.endswith('num_batches_tracked'):
        return torch.zeros((), dtype=torch.long)
    rs = np.random.RandomState(zlib.crc32(key.encode()) & 0x7fffffff)
    g = rs.standard_normal(shape).astype(np.float32) if len(shape) else np.float32(rs.standard_normal())
    if len(shape) >= 2:  # conv / linear weight: N(0, 2/fan_out)
        fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
        v = g * np.float32(np.sqrt(2.0 / fan_out))
    elif key.endswith('running_var'):
        v = 1.0 + 0.1 * np.abs(g)
    elif key.endswith('running_mean') or key.endswith('.bias'):
        v = 0.1 * g
    else:  # 1-D *.weight (SubBN affine)
        v = 1.0 + 0.1 * g
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def procedural_fill(keys):
    """state dict (key -> tensor) for a (key -> shape) table."""
    return OrderedDict((k, procedural_value(k, s)) for k, s in keys.items())


def fill_module_(module):
    """Overwrite every entry of ``module.state_dict()`` procedurally (works for
    the reference modules and for the product modules: same key names)."""
    sd = module.state_dict()
    new = OrderedDict((k, procedural_value(k, tuple(v.shape))) for k, v in sd.items())
    module.load_state_dict(new)
    return module


def rand_input(seed, shape, nonneg=False):
    """Inputs from the legacy numpy generator (stable across numpy versions)."""
    a = np.random.RandomState(seed).standard_normal(shape).astype(np.float32)
    if nonneg:
        a = np.abs(a)
    return torch.from_numpy(a)
