import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'coarse-fine-networks_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (HIP kernels through the C ABI)')
    config.addinivalue_line('markers', 'capture: captures a hipGraph (the library refuses capture in deterministic mode: skipped under CFN_DETERMINISTIC=1)')


def pytest_collection_modifyitems(config, items):
    if os.environ.get('CFN_DETERMINISTIC', '0') == '1':      # a whole-suite parity run in deterministic mode: graph capture is refused by design
        nocap = pytest.mark.skip(reason='deterministic mode cannot run inside a stream capture (by design)')
        for it in items:
            if 'capture' in it.keywords:
                it.add_marker(nocap)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_sd(z, field='keys'):
    """procedurally filled state dict for the key/shape list stored with a fixture"""
    from oracle import spec
    keys = json.loads(str(z[field]))
    return spec.procedural_fill({k: tuple(s) for k, s in keys})


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def maxdiff(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max())


def relerr(a, b):
    """max |a-b| / max |b|  (scale-normalised, for gradients of very different magnitude)"""
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
