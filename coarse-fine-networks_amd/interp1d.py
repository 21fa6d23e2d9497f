"""Interp1d -- batched 1-D linear interpolation on MI355X (drop-in for the reference's ``interp1d.py``).

``Interp1d()(x, y, xnew, out=None)`` keeps the reference call convention (interp1d.py:5-6: the object is
called like a function) and its shape rules (interp1d.py:20-71: 1-D or 2-D inputs, a single row
broadcasts against several).  The arithmetic runs in one HIP kernel (``cfn_interp1d_fwd``): binary
search = ``searchsorted`` (left) - 1 clamped to [0, N-2], then ``y0 + (y1-y0)/(eps+x1-x0)*(xnew-x0)``
with FMA contraction disabled, so both the int64 indices and the interpolated values are bit-identical
to the CPU reference (tests/test_hip_ops.py::test_interp1d_bit_exact).

Gradients w.r.t. x, y and xnew are provided by a second kernel with the index held constant -- the
behaviour the reference gets from plain autograd, since its custom ``backward`` is dead code
(SURVEY 3.4)."""
import sys

import torch

from cfn_hip import ops


class Interp1d(object):
    def __call__(self, x, y, xnew, out=None):
        return self.forward(x, y, xnew, out)

    @staticmethod
    def _rows(v, name):
        assert v.dim() <= 2, 'interp1d: all inputs must be at most 2-D.'
        return v[None, :] if v.dim() == 1 else v

    def forward(self, x, y, xnew, out=None, return_index=False):
        X, Y, Q = self._rows(x, 'x'), self._rows(y, 'y'), self._rows(xnew, 'xnew')
        assert len({str(t.device) for t in (X, Y, Q)}) == 1, 'All parameters must be on the same device.'
        assert X.shape[1] == Y.shape[1] and (X.shape[0] == Y.shape[0] or X.shape[0] == 1 or Y.shape[0] == 1), \
            'x and y must have the same number of columns, and either the same number of row or one of them having only one row.'
        stacked = X.shape[0] == 1 and Y.shape[0] == 1 and Q.shape[0] > 1      # interp1d.py:63-71
        qshape = Q.shape
        if stacked:
            Q = Q.contiguous().view(1, -1)
        o = ops
        fine = sys.modules.get('x3d_fine')
        if fine is not None and fine.USE_TORCH_OPS:       # CFN_USE_TORCH_OPS route: the registered dispatcher operator
            from cfn_hip import torchlib
            o = torchlib.TorchOps
        ynew, ind = o.interp1d(X.float(), Y.float(), Q.float())
        if stacked:
            ynew, ind = ynew.view(qshape), ind.view(qshape)
        if out is not None and out.numel() == ynew.numel():
            out.reshape(ynew.shape).copy_(ynew.detach())
        if return_index:
            return ynew, ind
        return ynew
