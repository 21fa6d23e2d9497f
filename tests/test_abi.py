"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU, exports every symbol
include/cfn_hip.h declares, and the product ops refuse to run without device tensors."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, PKG


def _lib():
    import cfn_hip
    if not os.path.exists(cfn_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return cfn_hip


def test_header_symbols_are_exported():
    cfn_hip = _lib()
    protos = cfn_hip.header_prototypes()
    assert len(protos) >= 30
    lib = ctypes.CDLL(cfn_hip.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), name
    # every extern "C" entry point in csrc is declared in the header (no undocumented ABI)
    declared = set(protos)
    for f in os.listdir(os.path.join(PKG, 'csrc')):
        if f.endswith('.hip'):
            src = open(os.path.join(PKG, 'csrc', f)).read()
            for m in re.finditer(r'extern\s+"C"\s+[\w\s\*]+?\b(cfn_\w+)\s*\(', src):
                assert m.group(1) in declared, (f, m.group(1))


def test_load_and_error_reporting_without_gpu():
    cfn_hip = _lib()
    lib = cfn_hip.load()
    assert lib.cfn_version().decode().startswith('cfn_hip')
    # argument validation happens before any launch: a null tensor is reported, not crashed on
    rc = lib.cfn_dwconv3d_fwd(None, None, None, 0, None, None, None, None, 1, 1, 1, 7, 7, 1, None)
    assert rc == 1 and 'null' in cfn_hip.last_error()
    rc = lib.cfn_pwconv_fwd(None, None, None, 0, None, None, None, None, 1, 1, 1, 1, 1, 1, 3, None)
    assert rc == 1


def test_ops_have_no_cpu_fallback():
    _lib()
    from cfn_hip import ops
    with pytest.raises(RuntimeError):
        ops.dwconv3d(torch.zeros(1, 2, 2, 7, 7), torch.zeros(2, 1, 3, 3, 3))
    with pytest.raises(RuntimeError):
        ops.time_sample(torch.zeros(1, 1, 4, 2, 2), torch.zeros(1, 3))


def test_product_modules_keep_reference_state_dict_keys():
    """drop-in contract: module attribute names = checkpoint keys"""
    import json
    import numpy as np
    import x3d_fine
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'state_keys.npz'))
    m = x3d_fine.generate_model('M', n_classes=400, task='loc', base_bn_splits=1)
    m.replace_logits(157)
    ref = {k: tuple(s) for k, s in json.loads(str(z['fine']))}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref
    m2 = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=2)
    ref2 = {k: tuple(s) for k, s in json.loads(str(z['fine_s2']))}
    assert {k: tuple(v.shape) for k, v in m2.state_dict().items()} == ref2
