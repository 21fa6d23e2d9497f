"""The plain-C restatement of the index arithmetic (oracle/index_ref.c) against the reference's golden vectors, the
torch-based oracle and ATen itself (CPU only; compiled by __graft_entry__.build_oracle with FMA contraction off)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def cref():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    so = ge.build_oracle()
    assert so and os.path.exists(so)
    lib = ctypes.CDLL(so)
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _grid_index(lib, cdf, T):
    cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    i0, w1 = np.empty(cdf.size, np.int32), np.empty(cdf.size, np.float32)
    lib.cfn_ref_grid_time_index(_p(cdf), ctypes.c_int(cdf.size), ctypes.c_int(T), _p(i0), _p(w1))
    return i0.reshape(cdf.shape), w1.reshape(cdf.shape)


def test_grid_time_index_on_reference_vectors(cref):
    for name, T in (('gridsample_ulp', 16), ('gridpool_d24_eval', 64), ('gridpool_d24_train', 64),
                    ('gridpool_d4_eval', 16), ('gridpool_d4_train', 16)):
        z = load_golden(name)
        i0, _ = _grid_index(cref, z['cdf'], T)
        assert np.array_equal(i0, z['i0'].astype(np.int32)), name


def test_grid_time_index_matches_torch_oracle(cref):
    from oracle import x3d_ref as R
    g = torch.Generator().manual_seed(5)
    for T in (16, 64, 256, 1000):
        p = torch.rand(32, 65, generator=g) + 0.05
        cdf = torch.cat([torch.zeros(32, 1), torch.cumsum((p / p.sum(1, keepdim=True)).double(), 1).float()], 1)
        i0c, w1c = R.grid_sample_time_index(cdf, T)
        i0, w1 = _grid_index(cref, cdf.numpy(), T)
        assert np.array_equal(i0, i0c.numpy()) and np.array_equal(w1, w1c.numpy())


@pytest.mark.parametrize('k', [5, 17, 65])
def test_interp1d_on_reference_vectors(cref, k):
    z = load_golden('interp1d_k%d' % k)
    from oracle import x3d_ref as R
    for xs, ys, qs, yref, iref in ((z['x'], z['mid'], z['mid'], z['ynew'], z['ind']), (z['x'], z['y2'], z['q2'], z['ynew2'], z['ind2'])):
        x, y, q = (np.ascontiguousarray(v, dtype=np.float32) for v in (xs, ys, qs))
        x, y, q = (v[None] if v.ndim == 1 else v for v in (x, y, q))
        B, N, P = max(x.shape[0], q.shape[0]), x.shape[1], q.shape[1]
        yn, ind = np.empty((B, P), np.float32), np.empty((B, P), np.int64)
        cref.cfn_ref_interp1d(_p(x), _p(y), _p(q), _p(yn), _p(ind), B, N, P, int(x.shape[0] > 1), int(y.shape[0] > 1),
                              int(q.shape[0] > 1))
        yo, io = R.interp1d(t(x), t(y), t(q))
        assert np.array_equal(ind, io.numpy().reshape(ind.shape)) and np.array_equal(yn, yo.numpy().reshape(yn.shape))
        assert np.array_equal(yn, np.asarray(yref, np.float32).reshape(yn.shape))          # the reference's own output
        assert np.array_equal(ind, np.asarray(iref).reshape(ind.shape))


@pytest.mark.parametrize('Tin,L,ac', [(16, 160, False), (64, 640, False), (64, 640, True), (17, 64, True), (9, 9, False)])
def test_resize_index_reproduces_aten(cref, Tin, L, ac):
    i0, i1, lam = np.empty(L, np.int32), np.empty(L, np.int32), np.empty(L, np.float32)
    cref.cfn_ref_resize_index(Tin, L, int(ac), _p(i0), _p(i1), _p(lam))
    x = torch.randn(3, 5, Tin, generator=torch.Generator().manual_seed(1))
    ref = F.interpolate(x, L, mode='linear', align_corners=ac).numpy()
    xn = x.numpy()
    out = (np.float32(1) - lam) * xn[:, :, i0] + lam * xn[:, :, i1]
    assert np.abs(out - ref).max() <= 1e-6
    assert i0.min() >= 0 and i1.max() <= Tin - 1 and np.all(np.diff(i0) >= 0)
