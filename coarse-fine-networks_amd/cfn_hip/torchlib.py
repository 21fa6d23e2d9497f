"""The hot-path kernels as registered torch operators (namespace ``cfn``): ``torch.ops.cfn.dwconv3d``, ``torch.ops.cfn.pwconv``,
``torch.ops.cfn.time_sample`` -- schemas, fake (meta) implementations and autograd formulas in the dispatcher, so that
``torch.compile`` / ``torch.export`` see the ops as opaque nodes instead of graph-breaking on ctypes calls.

    import cfn_hip.torchlib            # registers on import
    y, s, q = torch.ops.cfn.dwconv3d(x, w, A, B, act, stride)

The drop-in modules (x3d_fine / x3d_coarse) keep calling ``cfn_hip.ops`` directly: its autograd Functions carry the cross-op
fusions (shortcut tokens, tail links, batched gradient casts) that a functional op signature cannot express, and skip the
dispatcher's per-call cost (~1100 launches per step).  The operators here are the same C-ABI entry points (include/cfn_hip.h)
with plain semantics:

* ``cfn::dwconv3d(x, w, A?, B?, act, stride) -> (y, sum, sumsq)``   depthwise 3x3x3, pad 1, stride (1,s,s), input read through
  the prologue act(A x + B); per-(n,c) fp64 statistics of y              (x3d_fine.py:89-97 conv3x3x3 + the BN that follows)
* ``cfn::pwconv(x, w, A?, B?, act, stride) -> (y, sum, sumsq)``     1x1x1 conv, same conventions      (x3d_fine.py:100-105)
* ``cfn::time_sample(x, cdf) -> out``                                Grid Pool / Unpool resampler      (x3d_coarse.py:394-416)

No CPU implementation is registered: calling them with CPU tensors fails in the dispatcher.
"""
from typing import Optional, Tuple

import torch

from . import call, call_try, check, ACT_NONE  # noqa: F401

_lib = 'cfn'


def _sfx(t):
    return '_bf16' if t.dtype == torch.bfloat16 else ''


def _f64(*shape, dev):
    return torch.zeros(*shape, dtype=torch.float64, device=dev)


def _c64(t):
    return None if t is None else t.double().contiguous()


# ---------------------------------------------------------------------------------------------------------------------------
# depthwise 3x3x3
# ---------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_lib + '::dwconv3d', mutates_args=(), device_types='cuda')
def dwconv3d(x: torch.Tensor, w: torch.Tensor, A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0,
             stride: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    x = check(x).contiguous()
    N, C, T, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(N, C, T, Ho, Wo, dtype=x.dtype, device=x.device)
    s, q = _f64(N, C, dev=x.device), _f64(N, C, dev=x.device)
    call('cfn_dwconv3d_fwd' + _sfx(x), x, _c64(A), _c64(B), act, w.reshape(C, 27).float().contiguous(), y, s, q, N, C, T, H, W, stride)
    return y, s, q


@dwconv3d.register_fake
def _(x, w, A=None, B=None, act=0, stride=1):
    N, C, T, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    return (x.new_empty(N, C, T, Ho, Wo), x.new_empty(N, C, dtype=torch.float64), x.new_empty(N, C, dtype=torch.float64))


@torch.library.custom_op(_lib + '::dwconv3d_backward', mutates_args=(), device_types='cuda')
def dwconv3d_backward(gy: torch.Tensor, gs: torch.Tensor, gq: torch.Tensor, x: torch.Tensor, w: torch.Tensor, y: torch.Tensor,
                      A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0,
                      stride: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (gx, gw, gA, gB); gA / gB are zeros (1 element) when there is no prologue"""
    N, C, T, H, W = x.shape
    sfx = _sfx(x)
    gy = gy.contiguous()
    w2 = w.reshape(C, 27).float().contiguous()
    A64, B64 = _c64(A), _c64(B)
    gx = torch.empty_like(x)
    gw = _f64(C, 27, dev=x.device)
    ab = _f64(2, N, C, dev=x.device) if A is not None else None
    a64, b64 = (ab[0], ab[1]) if ab is not None else (None, None)
    gs64, gq64 = _c64(gs), _c64(gq)
    name = 'cfn_dwconv3d_bwd_fused' if stride == 1 else 'cfn_dwconv3d_bwd_fused_s2'
    if not call_try(name + sfx, gy, y, gs64, gq64, w2, x, A64, B64, act, gx, a64, b64, gw, N, C, T, H, W):
        call('cfn_dwconv3d_bwd_data' + sfx, gy, y, gs64, gq64, w2, x, A64, B64, act, gx, a64, b64, N, C, T, H, W, stride)
        call('cfn_dwconv3d_bwd_weight' + sfx, gy, y, gs64, gq64, x, A64, B64, act, gw, N, C, T, H, W, stride)
    if ab is None:      # (outputs of a custom op may not alias each other)
        return gx, gw.float().view(w.shape), x.new_zeros(1, dtype=torch.float32), x.new_zeros(1, dtype=torch.float32)
    return gx, gw.float().view(w.shape), ab[0].float(), ab[1].float()


@dwconv3d_backward.register_fake
def _(gy, gs, gq, x, w, y, A=None, B=None, act=0, stride=1):
    n = (x.shape[0], x.shape[1]) if A is not None else (1,)
    return (torch.empty_like(x), torch.empty_like(w, dtype=torch.float32), x.new_empty(n, dtype=torch.float32),
            x.new_empty(n, dtype=torch.float32))


def _dw_setup(ctx, inputs, output):
    x, w, A, B, act, stride = inputs
    y, _, _ = output
    ctx.save_for_backward(x, w, y, A, B)
    ctx.act, ctx.stride = act, stride


def _dw_backward(ctx, gy, gs, gq):
    x, w, y, A, B = ctx.saved_tensors
    gy = torch.zeros_like(y) if gy is None else gy
    gs = torch.zeros(y.shape[:2], dtype=torch.float64, device=y.device) if gs is None else gs
    gq = torch.zeros(y.shape[:2], dtype=torch.float64, device=y.device) if gq is None else gq
    gx, gw, gA, gB = torch.ops.cfn.dwconv3d_backward(gy, gs, gq, x, w, y, A, B, ctx.act, ctx.stride)
    return gx, gw, (gA if A is not None else None), (gB if B is not None else None), None, None


dwconv3d.register_autograd(_dw_backward, setup_context=_dw_setup)


# ---------------------------------------------------------------------------------------------------------------------------
# pointwise 1x1x1
# ---------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_lib + '::pwconv', mutates_args=(), device_types='cuda')
def pwconv(x: torch.Tensor, w: torch.Tensor, A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0,
           stride: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    x = check(x).contiguous()
    if x.dtype != torch.float32:
        raise RuntimeError('cfn::pwconv: fp32 tensors (the bf16 pointwise path is reached through cfn_hip.ops)')
    N, Cin, T, H, W = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(N, Cout, T, Ho, Wo, dtype=x.dtype, device=x.device)
    s, q = _f64(N, Cout, dev=x.device), _f64(N, Cout, dev=x.device)
    call('cfn_pwconv_fwd', x, _c64(A), _c64(B), act, w.reshape(Cout, Cin).float().contiguous(), y, s, q, N, Cin, Cout, T, H, W, stride)
    return y, s, q


@pwconv.register_fake
def _(x, w, A=None, B=None, act=0, stride=1):
    N, Cin, T, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    return (x.new_empty(N, w.shape[0], T, Ho, Wo), x.new_empty(N, w.shape[0], dtype=torch.float64),
            x.new_empty(N, w.shape[0], dtype=torch.float64))


@torch.library.custom_op(_lib + '::pwconv_backward', mutates_args=(), device_types='cuda')
def pwconv_backward(gy: torch.Tensor, gs: torch.Tensor, gq: torch.Tensor, x: torch.Tensor, w: torch.Tensor, y: torch.Tensor,
                    A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0,
                    stride: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    N, Cin, T, H, W = x.shape
    Cout = w.shape[0]
    gy = gy.contiguous()
    w2 = w.reshape(Cout, Cin).float().contiguous()
    A64, B64, gs64, gq64 = _c64(A), _c64(B), _c64(gs), _c64(gq)
    gx = torch.zeros_like(x) if stride != 1 else torch.empty_like(x)
    gw = _f64(Cout, Cin, dev=x.device)
    ab = _f64(2, N, Cin, dev=x.device) if A is not None else None
    a64, b64 = (ab[0], ab[1]) if ab is not None else (None, None)
    call('cfn_pwconv_bwd_data', gy, y, gs64, gq64, w2, x, A64, B64, act, gx, a64, b64, N, Cin, Cout, T, H, W, stride)
    call('cfn_pwconv_bwd_weight', gy, y, gs64, gq64, x, A64, B64, act, gw, N, Cin, Cout, T, H, W, stride, None)
    if ab is None:      # (outputs of a custom op may not alias each other)
        return gx, gw.float().view(w.shape), x.new_zeros(1, dtype=torch.float32), x.new_zeros(1, dtype=torch.float32)
    return gx, gw.float().view(w.shape), ab[0].float(), ab[1].float()


@pwconv_backward.register_fake
def _(gy, gs, gq, x, w, y, A=None, B=None, act=0, stride=1):
    n = (x.shape[0], x.shape[1]) if A is not None else (1,)
    return (torch.empty_like(x), torch.empty_like(w, dtype=torch.float32), x.new_empty(n, dtype=torch.float32),
            x.new_empty(n, dtype=torch.float32))


def _pw_backward(ctx, gy, gs, gq):
    x, w, y, A, B = ctx.saved_tensors
    gy = torch.zeros_like(y) if gy is None else gy
    gs = torch.zeros(y.shape[:2], dtype=torch.float64, device=y.device) if gs is None else gs
    gq = torch.zeros(y.shape[:2], dtype=torch.float64, device=y.device) if gq is None else gq
    gx, gw, gA, gB = torch.ops.cfn.pwconv_backward(gy, gs, gq, x, w, y, A, B, ctx.act, ctx.stride)
    return gx, gw, (gA if A is not None else None), (gB if B is not None else None), None, None


pwconv.register_autograd(_pw_backward, setup_context=_dw_setup)


# ---------------------------------------------------------------------------------------------------------------------------
# Grid Pool / Grid Unpool resampler
# ---------------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(_lib + '::time_sample', mutates_args=(), device_types='cuda')
def time_sample(x: torch.Tensor, cdf: torch.Tensor) -> torch.Tensor:
    x, cdf = check(x).contiguous(), check(cdf).contiguous()
    B, C, Tin = x.shape[:3]
    K = cdf.shape[1]
    out = torch.empty((B, C, K) + tuple(x.shape[3:]), dtype=torch.float32, device=x.device)
    call('cfn_time_sample_fwd', x, cdf, out, B, C, Tin, K, x[0, 0, 0].numel())
    return out


@time_sample.register_fake
def _(x, cdf):
    return x.new_empty((x.shape[0], x.shape[1], cdf.shape[1]) + tuple(x.shape[3:]))


@torch.library.custom_op(_lib + '::time_sample_backward', mutates_args=(), device_types='cuda')
def time_sample_backward(g: torch.Tensor, x: torch.Tensor, cdf: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    B, C, Tin = x.shape[:3]
    K = cdf.shape[1]
    gx = torch.empty_like(x)
    g64 = _f64(B, K, dev=x.device)
    call('cfn_time_sample_bwd', g.contiguous(), x.contiguous(), cdf.contiguous(), gx, g64, B, C, Tin, K, x[0, 0, 0].numel())
    return gx, g64.float()


@time_sample_backward.register_fake
def _(g, x, cdf):
    return torch.empty_like(x), torch.empty_like(cdf)


def _ts_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _ts_backward(ctx, g):
    x, cdf = ctx.saved_tensors
    return torch.ops.cfn.time_sample_backward(g, x, cdf)


time_sample.register_autograd(_ts_backward, setup_context=_ts_setup)
