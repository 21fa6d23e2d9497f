"""Deterministic parameters / inputs for the tools (the tools must not import the test-only oracle package)."""
import zlib

import numpy as np
import torch


def fill_module_(module):
    """overwrite every state_dict entry procedurally: conv / linear weights ~ N(0, 2 / fan_in), norm weights 1 +- 0.1, biases and running means
    +- 0.1, running variances in [0.5, 1.5]; seeded by the entry's key"""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
        if k.endswith('num_batches_tracked'):
            new[k] = torch.zeros_like(v)
        elif k.endswith('running_var'):
            new[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            new[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith('weight'):
            new[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            new[k] = 0.1 * torch.randn(v.shape, generator=g)
    module.load_state_dict(new)
    return module


def rand_input(seed, shape, nonneg=False):
    a = np.random.RandomState(seed).standard_normal(shape).astype(np.float32)
    return torch.from_numpy(np.abs(a) if nonneg else a)
