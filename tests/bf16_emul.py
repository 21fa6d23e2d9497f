"""bf16 emulation of the CPU oracle (test infrastructure): every convolution output, every block output and every
gradient flowing back through them is rounded to bf16 (round to nearest even) -- bf16 STORAGE -- and the two operands of
every pointwise (1x1x1) convolution are rounded to bf16 before the fp32-accumulated product -- bf16 MFMA operands; all
other arithmetic stays fp32.  This is the accuracy a bf16 activation path with bf16 matrix cores has by construction; the
GPU tests require the HIP bf16 path to stay within a small factor of this floor when both are compared with the fp32
reference."""
import contextlib

import torch

from oracle import x3d_ref as R


class RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


class RoundFwd(torch.autograd.Function):
    """operand rounding: the value is rounded, the gradient passes (the products' accumulators are fp32)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


@contextlib.contextmanager
def bf16_storage(mfma_operands=True):
    conv, block = R.F.conv3d, R.bottleneck

    def conv_q(inp, weight, *a, **k):
        if mfma_operands and tuple(weight.shape[2:]) == (1, 1, 1) and k.get('groups', 1) == 1:
            inp, weight = RoundFwd.apply(inp), RoundFwd.apply(weight)
        return RoundBf16.apply(conv(inp, weight, *a, **k))

    R.F.conv3d = conv_q
    R.bottleneck = lambda *a, **k: RoundBf16.apply(block(*a, **k))
    try:
        yield
    finally:
        R.F.conv3d, R.bottleneck = conv, block


def nrel(a, b):
    """||a - b|| / ||b||"""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
