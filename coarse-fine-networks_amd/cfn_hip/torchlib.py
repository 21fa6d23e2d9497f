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

import os

from . import call, call_try, check, load, ACT_NONE  # noqa: F401

_lib = 'cfn'

# Native registration (csrc/torch/cfn_torch.cpp -> cfn_hip/libcfn_torch.so, built by __graft_entry__.build()): ALL 16 forward + 16 backward
# operators of the set are defined and implemented by TORCH_LIBRARY inside a shared library (round 5: dwconv3d / pwconv /
# time_sample, round 6: the other 13 pairs).  When it is present the Python definitions below are skipped; their fake implementations and
# autograd formulas are attached to the native operators.  CFN_NATIVE_OPS=0: Python custom_op definitions over ctypes for everything (the
# round-3/4 route; also what serves 16-bit tensors through bn_add_relu / pool_hw / dwconv_t5 -- the native operators are fp32).
NATIVE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcfn_torch.so')
NATIVE = False
if os.environ.get('CFN_NATIVE_OPS', '1') != '0' and os.path.exists(NATIVE_LIB):
    load()                                   # libcfn_hip.so first: the native library is linked against it
    torch.ops.load_library(NATIVE_LIB)
    NATIVE = True


class _NativeOp(object):
    """what torch.library.custom_op returns, for an operator that a TORCH_LIBRARY block already defined"""

    def __init__(self, qualname):
        self.qualname = qualname
        ns, name = qualname.split('::')
        self._op = getattr(getattr(torch.ops, ns), name)

    def __call__(self, *a, **k):
        return self._op(*a, **k)

    def register_fake(self, fn):
        torch.library.register_fake(self.qualname)(fn)
        return fn

    def register_autograd(self, backward, setup_context=None):
        torch.library.register_autograd(self.qualname, backward, setup_context=setup_context)


def _custom_or_native(qualname, **kw):
    def deco(fn):
        if NATIVE:
            return _NativeOp(qualname)
        return torch.library.custom_op(qualname, **kw)(fn)
    return deco


def _sfx(t):
    return '_bf16' if t.dtype == torch.bfloat16 else ('_f16' if t.dtype == torch.float16 else '')


def _f64(*shape, dev):
    return torch.zeros(*shape, dtype=torch.float64, device=dev)


def _c64(t):
    return None if t is None else t.double().contiguous()


# ---------------------------------------------------------------------------------------------------------------------------
# depthwise 3x3x3
# ---------------------------------------------------------------------------------------------------------------------------
@_custom_or_native(_lib + '::dwconv3d', mutates_args=(), device_types='cuda')
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


@_custom_or_native(_lib + '::dwconv3d_backward', mutates_args=(), device_types='cuda')
def dwconv3d_backward(gy: torch.Tensor, gs: torch.Tensor, gq: torch.Tensor, x: torch.Tensor, w: torch.Tensor, y: torch.Tensor,
                      A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0,
                      stride: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (gx, gw, gA, gB); gA / gB are zeros (1 element) when there is no prologue"""
    x, y = x.contiguous(), y.contiguous()      # the forward computed on a contiguous copy; the saved tensor is the caller's
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
@_custom_or_native(_lib + '::pwconv', mutates_args=(), device_types='cuda')
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


@_custom_or_native(_lib + '::pwconv_backward', mutates_args=(), device_types='cuda')
def pwconv_backward(gy: torch.Tensor, gs: torch.Tensor, gq: torch.Tensor, x: torch.Tensor, w: torch.Tensor, y: torch.Tensor,
                    A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0,
                    stride: int = 1) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    x, y = x.contiguous(), y.contiguous()
    N, Cin, T, H, W = x.shape
    Cout = w.shape[0]
    gy = gy.contiguous()
    w2 = w.reshape(Cout, Cin).float().contiguous()
    A64, B64, gs64, gq64 = _c64(A), _c64(B), _c64(gs), _c64(gq)
    gx = torch.zeros_like(x) if stride != 1 else torch.empty_like(x)
    gw = _f64(Cout, Cin, dev=x.device)
    ab = _f64(2, N, Cin, dev=x.device) if A is not None else None
    a64, b64 = (ab[0], ab[1]) if ab is not None else (None, None)
    # one pass over gy, y, x where the library has a fused kernel for the shape (layers 1 and 2: pwfused.hip, pwfuseds.hip), as cfn_hip.ops does
    if not (stride == 1 and call_try('cfn_pwconv_bwd_fused', gy, y, gs64, gq64, w2, x, A64, B64, act, gx, a64, b64, gw, N, Cin, Cout, T, H, W, None, 1, None)):
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
@_custom_or_native(_lib + '::time_sample', mutates_args=(), device_types='cuda')
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


@_custom_or_native(_lib + '::time_sample_backward', mutates_args=(), device_types='cuda')
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


# ---------------------------------------------------------------------------------------------------------------------------
# The rest of the section-8(b) operator set.  Each operator is the SAME code path as the autograd Function of cfn_hip.ops
# (its forward / backward bodies are called with a plain context object), wrapped in a dispatcher schema + fake (meta)
# implementation + registered autograd formula.  Plain semantics: no ShortcutToken / TailLink / lazily cast gradients.
# Coefficients (A, B) may be fp32 or fp64 (the C ABI takes fp64); their gradients come back in the dtype of the input.
# ---------------------------------------------------------------------------------------------------------------------------
from typing import List  # noqa: E402

from . import ops as _ops  # noqa: E402


class _Ctx(object):
    """what the Function bodies of cfn_hip.ops need from an autograd context"""

    def __init__(self, saved=(), **attrs):
        self.saved_tensors = tuple(saved)
        self.needs_input_grad = (True,) * 16
        for k, v in attrs.items():
            setattr(self, k, v)

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *ts):
        pass


def _z(ref, dtype=torch.float32):
    """placeholder for "no tensor" in an operator result (results may not be None)"""
    return ref.new_zeros(1, dtype=dtype)


def _own(t, like=None, dtype=None):
    """a result tensor that aliases nothing (scratch-arena views, inputs) in the dtype of `like`"""
    if t is None:
        return None
    dt = dtype if dtype is not None else (like.dtype if like is not None else t.dtype)
    return t.to(dt).clone() if t.dtype == dt else t.to(dt)


def _orz(t, ref, like=None):
    return _z(ref) if t is None else _own(t, like)


def _gz(g, like, dtype=None):
    """incoming gradient or zeros (autograd passes None for unused outputs)"""
    return torch.zeros_like(like, dtype=dtype or like.dtype) if g is None else g


def _op(name, **kw):
    """round 6: these 26 operators are native too (csrc/torch/cfn_torch.cpp); the Python bodies below are what CFN_NATIVE_OPS=0 registers"""
    return _custom_or_native(_lib + '::' + name, mutates_args=kw.pop('mutates_args', ()), device_types='cuda', **kw)


# ---- conv1_t: depthwise 5x1x1 ------------------------------------------------------------------------------------------------
@_op('dwconv_t5')
def dwconv_t5(x: torch.Tensor, w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    y, s, q = _ops._DwConvT5.forward(_Ctx(), x, w, True, None)
    return y, s.clone(), q.clone()


@dwconv_t5.register_fake
def _(x, w):
    return torch.empty_like(x), x.new_empty(x.shape[:2], dtype=torch.float64), x.new_empty(x.shape[:2], dtype=torch.float64)


@_op('dwconv_t5_backward')
def dwconv_t5_backward(gy: torch.Tensor, gs: torch.Tensor, gq: torch.Tensor, x: torch.Tensor, w: torch.Tensor,
                       y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    C = x.shape[1]
    c = _Ctx((x.contiguous(), w.reshape(C, 5).float().contiguous(), y.contiguous()), wparam=w)
    gx, gw = _ops._DwConvT5.backward(c, gy, gs.double(), gq.double())[:2]
    return gx, _own(gw, dtype=torch.float32).view(w.shape)


@dwconv_t5_backward.register_fake
def _(gy, gs, gq, x, w, y):
    return torch.empty_like(x), torch.empty_like(w, dtype=torch.float32)


def _t5_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[0])


def _t5_backward(ctx, gy, gs, gq):
    x, w, y = ctx.saved_tensors
    f64 = lambda g: torch.zeros(y.shape[:2], dtype=torch.float64, device=y.device) if g is None else g
    return torch.ops.cfn.dwconv_t5_backward(_gz(gy, y), f64(gs), f64(gq), x, w, y)


dwconv_t5.register_autograd(_t5_backward, setup_context=_t5_setup)


# ---- conv1_s: dense 1x3x3 stem conv (the clip gets no gradient) ----------------------------------------------------------------
@_op('stem_conv')
def stem_conv(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return _ops._StemConv.forward(_Ctx(), x, w)


@stem_conv.register_fake
def _(x, w):
    N, _, T, H, W = x.shape
    return x.new_empty(N, w.shape[0], T, (H - 1) // 2 + 1, (W - 1) // 2 + 1)


@_op('stem_conv_backward')
def stem_conv_backward(gy: torch.Tensor, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    c = _Ctx((x.contiguous(),), wshape=tuple(w.shape), wparam=w)
    return _own(_ops._StemConv.backward(c, gy)[1], dtype=torch.float32).view(w.shape)


@stem_conv_backward.register_fake
def _(gy, x, w):
    return torch.empty_like(w, dtype=torch.float32)


stem_conv.register_autograd(lambda ctx, gy: (None, torch.ops.cfn.stem_conv_backward(gy, *ctx.saved_tensors)),
                            setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


# ---- dense conv3d (Grid Pool saliency convs) -------------------------------------------------------------------------------------
@_op('conv3d_dense')
def conv3d_dense(x: torch.Tensor, w: torch.Tensor, kernel: List[int], stride: List[int], padding: List[int],
                 A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None,
                 act: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    y, s, q = _ops._ConvDense.forward(_Ctx(), x, A, B, w, act, tuple(kernel), tuple(stride), tuple(padding), True)
    return y, s.clone(), q.clone()


@conv3d_dense.register_fake
def _(x, w, kernel, stride, padding, A=None, B=None, act=0):
    N, _, T, H, W = x.shape
    o = [(d + 2 * p - k) // s + 1 for d, k, s, p in zip((T, H, W), kernel, stride, padding)]
    return (x.new_empty(N, w.shape[0], *o), x.new_empty(N, w.shape[0], dtype=torch.float64), x.new_empty(N, w.shape[0], dtype=torch.float64))


@_op('conv3d_dense_backward')
def conv3d_dense_backward(gy: torch.Tensor, gs: torch.Tensor, gq: torch.Tensor, x: torch.Tensor, w: torch.Tensor, y: torch.Tensor,
                          kernel: List[int], stride: List[int], padding: List[int], A: Optional[torch.Tensor] = None,
                          B: Optional[torch.Tensor] = None,
                          act: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    c = _Ctx((x.contiguous(), _c64(A), _c64(B), w.reshape(w.shape[0], -1).float().contiguous(), y.contiguous()),
             meta=(act, tuple(kernel), tuple(stride), tuple(padding), tuple(w.shape)), wparam=w)
    gx, gA, gB, gw = _ops._ConvDense.backward(c, gy, gs.double(), gq.double())[:4]
    return gx, _own(gw, dtype=torch.float32).view(w.shape), _orz(gA, x, A), _orz(gB, x, B)


@conv3d_dense_backward.register_fake
def _(gy, gs, gq, x, w, y, kernel, stride, padding, A=None, B=None, act=0):
    ab = (lambda t: torch.empty_like(t)) if A is not None else (lambda t: x.new_empty(1))
    return torch.empty_like(x), torch.empty_like(w, dtype=torch.float32), ab(A), ab(B)


def _cd_setup(ctx, inputs, output):
    x, w, kernel, stride, padding, A, B, act = inputs
    ctx.save_for_backward(x, w, output[0], A, B)
    ctx.geom = (kernel, stride, padding, act)


def _cd_backward(ctx, gy, gs, gq):
    x, w, y, A, B = ctx.saved_tensors
    kernel, stride, padding, act = ctx.geom
    f64 = lambda g: torch.zeros(y.shape[:2], dtype=torch.float64, device=y.device) if g is None else g
    gx, gw, gA, gB = torch.ops.cfn.conv3d_dense_backward(_gz(gy, y), f64(gs), f64(gq), x, w, y, kernel, stride, padding, A, B, act)
    return gx, gw, None, None, None, (gA if A is not None else None), (gB if B is not None else None), None


conv3d_dense.register_autograd(_cd_backward, setup_context=_cd_setup)


# ---- SubBatchNorm3d statistics -> prologue coefficients (+ SE gate) ---------------------------------------------------------------
@_op('bn_fold')
def bn_fold(s: Optional[torch.Tensor], q: Optional[torch.Tensor], gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
            run_mean: torch.Tensor, run_var: torch.Tensor, nbt: torch.Tensor, training: bool, N: int, C: int, S: int, count: float,
            eps: float, momentum: float, w1: Optional[torch.Tensor] = None, b1: Optional[torch.Tensor] = None,
            w2: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None,
            pool_count: float = 1.0) -> List[torch.Tensor]:
    """-> [A, B, mean, rstd, A0, B0, gate, hbuf, pooled, new_run_mean, new_run_var, new_nbt].  FUNCTIONAL: the running statistics
    are returned, not updated in place (a dispatcher operator with an autograd formula may not mutate its arguments); the caller
    copies them into its buffers (x3d_fine.SubBatchNorm3d.fold does).  A0 .. pooled are 1-element placeholders without an SE branch."""
    c = _Ctx()
    rm, rv, nb = run_mean.clone(), run_var.clone(), nbt.clone()
    A, B = _ops._BnFold.forward(c, s, q, gamma, beta, w1, b1, w2, b2, (rm, rv, nb), (training, N, C, S, count, eps, momentum, pool_count))
    _s, _g, mean, rstd, A0, B0, gate, hbuf, pooled, _w1, _w2 = c.saved_tensors
    return [A, B, mean, rstd] + [(_z(A) if t is None else t) for t in (A0, B0, gate, hbuf, pooled)] + [rm, rv, nb]


@bn_fold.register_fake
def _(s, q, gamma, beta, run_mean, run_var, nbt, training, N, C, S, count, eps, momentum, w1=None, b1=None, w2=None, b2=None,
      pool_count=1.0):
    dev = run_mean.device
    f64 = lambda *sh: torch.empty(*sh, dtype=torch.float64, device=dev)
    f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
    Se = S if training else 1
    new = [torch.empty_like(run_mean), torch.empty_like(run_var), torch.empty_like(nbt)]
    if w1 is None:
        return [f64(N, C), f64(N, C), f64(Se, C), f64(Se, C)] + [f32(1) for _ in range(5)] + new
    return [f64(N, C), f64(N, C), f64(Se, C), f64(Se, C), f32(N, C), f32(N, C), f32(N, C), f32(N, w1.shape[0]), f32(N, C)] + new


@_op('bn_fold_backward')
def bn_fold_backward(gA: torch.Tensor, gB: torch.Tensor, s: Optional[torch.Tensor], gamma: Optional[torch.Tensor], saved: List[torch.Tensor],
                     training: bool, N: int, C: int, S: int, count: float, pool_count: float, w1: Optional[torch.Tensor] = None,
                     w2: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """-> [gs, gq, ggamma, gbeta, gw1, gb1, gw2, gb2] (1-element placeholders where there is nothing)"""
    mean, rstd, A0, B0, gate, hbuf, pooled = saved
    Wd = w1.shape[0] if w1 is not None else 0
    none = lambda t: t if Wd else None
    c = _Ctx((None if s is None else s.double().contiguous(), gamma, mean, rstd, none(A0), none(B0), none(gate), none(hbuf), none(pooled),
              None if w1 is None else w1.reshape(Wd, C).contiguous(), None if w2 is None else w2.reshape(C, Wd).contiguous()),
             cfg=(training, N, C, S, Wd, count, pool_count, None if w1 is None else tuple(w1.shape), None if w2 is None else tuple(w2.shape)))
    outs = _ops._BnFold.backward(c, gA.double(), gB.double())[:8]
    return [_z(mean) if t is None else t.clone() for t in outs]


@bn_fold_backward.register_fake
def _(gA, gB, s, gamma, saved, training, N, C, S, count, pool_count, w1=None, w2=None):
    dev = gA.device
    e = lambda *sh, dt=torch.float32: torch.empty(*sh, dtype=dt, device=dev)
    has_s = training or w1 is not None
    out = [e(N, C, dt=torch.float64) if has_s else e(1), e(N, C, dt=torch.float64) if training else e(1)]
    out += [e(C), e(C)] if gamma is not None else [e(1), e(1)]
    if w1 is not None:
        out += [torch.empty_like(w1, dtype=torch.float32), e(w1.shape[0]), torch.empty_like(w2, dtype=torch.float32), e(C)]
    else:
        out += [e(1) for _ in range(4)]
    return out


def _bf_setup(ctx, inputs, output):
    (s, q, gamma, beta, run_mean, run_var, nbt, training, N, C, S, count, eps, momentum, w1, b1, w2, b2, pool_count) = inputs
    ctx.save_for_backward(s, gamma, w1, w2, *output[2:9])
    ctx.cfg = (training, N, C, S, count, pool_count, q is not None, beta is not None, b1 is not None, b2 is not None)


def _bf_backward(ctx, grads):
    s, gamma, w1, w2 = ctx.saved_tensors[:4]
    saved = list(ctx.saved_tensors[4:])
    training, N, C, S, count, pool_count, has_q, has_beta, has_b1, has_b2 = ctx.cfg
    z = lambda: torch.zeros(N, C, dtype=torch.float64, device=saved[0].device)
    gA = z() if grads[0] is None else grads[0]
    gB = z() if grads[1] is None else grads[1]
    gs, gq, gg, gbt, gw1, gb1, gw2, gb2 = torch.ops.cfn.bn_fold_backward(gA, gB, s, gamma, saved, training, N, C, S, count, pool_count, w1, w2)
    has_s = s is not None and (training or w1 is not None)
    return (gs if has_s else None, gq if (training and has_q) else None, gg if gamma is not None else None,
            gbt if has_beta else None, None, None, None, None, None, None, None, None, None, None,
            gw1 if w1 is not None else None, gb1 if has_b1 else None, gw2 if w2 is not None else None, gb2 if has_b2 else None, None)


bn_fold.register_autograd(_bf_backward, setup_context=_bf_setup)


# ---- block tail: relu(A y + B + (Ar res + Br)) -------------------------------------------------------------------------------------
@_op('bn_add_relu')
def bn_add_relu(y: torch.Tensor, A: torch.Tensor, B: torch.Tensor, res: torch.Tensor, Ar: Optional[torch.Tensor] = None,
                Br: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _ops._BnAddRelu.forward(_Ctx(), y, A, B, res, Ar, Br, False, None)


@bn_add_relu.register_fake
def _(y, A, B, res, Ar=None, Br=None):
    return torch.empty_like(y)


@_op('bn_add_relu_backward')
def bn_add_relu_backward(gout: torch.Tensor, y: torch.Tensor, A: torch.Tensor, res: torch.Tensor, out: torch.Tensor,
                         Ar: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """-> [gy, gA, gB, gres, gAr, gBr]"""
    c = _Ctx((y.contiguous(), _c64(A), res.contiguous(), _c64(Ar), out.contiguous()), link=None, has_mask=False)
    gy, gA, gB, gres, gAr, gBr = _ops._BnAddRelu.backward(c, gout)[:6]
    return [gy, _own(gA, A), _own(gB, A), gres, _orz(gAr, y, Ar), _orz(gBr, y, Ar)]


@bn_add_relu_backward.register_fake
def _(gout, y, A, res, out, Ar=None):
    r = (lambda: torch.empty_like(Ar)) if Ar is not None else (lambda: y.new_empty(1))
    return [torch.empty_like(y), torch.empty_like(A), torch.empty_like(A), torch.empty_like(res), r(), r()]


def _bar_setup(ctx, inputs, output):
    y, A, B, res, Ar, Br = inputs
    ctx.save_for_backward(y, A, res, Ar, output)
    ctx.has_br = Br is not None


def _bar_backward(ctx, gout):
    y, A, res, Ar, out = ctx.saved_tensors
    gy, gA, gB, gres, gAr, gBr = torch.ops.cfn.bn_add_relu_backward(gout, y, A, res, out, Ar)
    return gy, gA, gB, gres, (gAr if Ar is not None else None), (gBr if ctx.has_br else None)


bn_add_relu.register_autograd(_bar_backward, setup_context=_bar_setup)


# ---- act(A x + B) materialised ---------------------------------------------------------------------------------------------------
@_op('affine_act')
def affine_act(x: torch.Tensor, A: torch.Tensor, B: torch.Tensor, act: int = 0) -> torch.Tensor:
    return _ops._AffineAct.forward(_Ctx(), x, A, B, act)


@affine_act.register_fake
def _(x, A, B, act=0):
    return torch.empty_like(x)


@_op('affine_act_backward')
def affine_act_backward(gout: torch.Tensor, x: torch.Tensor, A: torch.Tensor, B: torch.Tensor, act: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    gx, gA, gB = _ops._AffineAct.backward(_Ctx((x.contiguous(), _c64(A), _c64(B)), act=act), gout)[:3]
    return gx, _own(gA, A), _own(gB, B)


@affine_act_backward.register_fake
def _(gout, x, A, B, act):
    return torch.empty_like(x), torch.empty_like(A), torch.empty_like(B)


def _aa_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:3])
    ctx.act = inputs[3]


affine_act.register_autograd(lambda ctx, g: (*torch.ops.cfn.affine_act_backward(g, *ctx.saved_tensors, ctx.act), None), setup_context=_aa_setup)


# ---- adaptive (OH, OW) spatial mean of act(A x + B) ------------------------------------------------------------------------------
@_op('pool_hw')
def pool_hw(x: torch.Tensor, OH: int, OW: int, A: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None, act: int = 0) -> torch.Tensor:
    return _ops._PoolHW.forward(_Ctx(), x, A, B, act, OH, OW)


@pool_hw.register_fake
def _(x, OH, OW, A=None, B=None, act=0):
    return x.new_empty(tuple(x.shape[:3]) + (OH, OW), dtype=torch.float32)


@_op('pool_hw_backward')
def pool_hw_backward(gout: torch.Tensor, x: torch.Tensor, OH: int, OW: int, A: Optional[torch.Tensor] = None,
                     B: Optional[torch.Tensor] = None, act: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    gx, gA, gB = _ops._PoolHW.backward(_Ctx((x.contiguous(), _c64(A), _c64(B)), meta=(act, OH, OW)), gout)[:3]
    return gx, _orz(gA, x, A), _orz(gB, x, B)


@pool_hw_backward.register_fake
def _(gout, x, OH, OW, A=None, B=None, act=0):
    ab = (lambda t: torch.empty_like(t)) if A is not None else (lambda t: x.new_empty(1, dtype=torch.float32))
    return torch.empty_like(x), ab(A), ab(B)


def _ph_setup(ctx, inputs, output):
    x, OH, OW, A, B, act = inputs
    ctx.save_for_backward(x, A, B)
    ctx.meta = (OH, OW, act)


def _ph_backward(ctx, g):
    x, A, B = ctx.saved_tensors
    OH, OW, act = ctx.meta
    gx, gA, gB = torch.ops.cfn.pool_hw_backward(g, x, OH, OW, A, B, act)
    return gx, None, None, (gA if A is not None else None), (gB if B is not None else None), None


pool_hw.register_autograd(_ph_backward, setup_context=_ph_setup)


# ---- Interp1d (interp1d.py:8-147) ---------------------------------------------------------------------------------------------------
@_op('interp1d')
def interp1d(x: torch.Tensor, y: torch.Tensor, xnew: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return _ops._Interp1d.forward(_Ctx(), x, y, xnew)


@interp1d.register_fake
def _(x, y, xnew):
    B = max(x.shape[0], y.shape[0], xnew.shape[0])
    return x.new_empty(B, xnew.shape[1], dtype=torch.float32), x.new_empty(B, xnew.shape[1], dtype=torch.int64)


@_op('interp1d_backward')
def interp1d_backward(g: torch.Tensor, x: torch.Tensor, y: torch.Tensor, xnew: torch.Tensor,
                      ind: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    rows = (int(x.shape[0] > 1), int(y.shape[0] > 1), int(xnew.shape[0] > 1))
    return _ops._Interp1d.backward(_Ctx((x.contiguous(), y.contiguous(), xnew.contiguous(), ind), rows=rows), g, None)


@interp1d_backward.register_fake
def _(g, x, y, xnew, ind):
    return torch.empty_like(x), torch.empty_like(y), torch.empty_like(xnew)


interp1d.register_autograd(lambda ctx, g, _gi: torch.ops.cfn.interp1d_backward(_gz(g, ctx.saved_tensors[3], torch.float32), *ctx.saved_tensors),
                           setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs, output[1]))


# ---- Grid Pool CDF (x3d_coarse.py:384-392) ---------------------------------------------------------------------------------------------
@_op('grid_cdf')
def grid_cdf(g: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _ops._GridCdf.forward(_Ctx(), g, bias)


@grid_cdf.register_fake
def _(g, bias=None):
    return g.new_empty(g.shape[0], g.shape[1] + 1)


@_op('grid_cdf_backward')
def grid_cdf_backward(gcdf: torch.Tensor, g: torch.Tensor, bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    gg, gb = _ops._GridCdf.backward(_Ctx((g.contiguous(), bias)), gcdf)
    return gg, _orz(gb, g)


@grid_cdf_backward.register_fake
def _(gcdf, g, bias=None):
    return torch.empty_like(g), (torch.empty_like(bias) if bias is not None else g.new_empty(1))


def _gc_backward(ctx, gcdf):
    g, bias = ctx.saved_tensors
    gg, gb = torch.ops.cfn.grid_cdf_backward(gcdf, g, bias)
    return gg, (gb if bias is not None else None)


grid_cdf.register_autograd(_gc_backward, setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


# ---- Gaussian temporal alignment (x3d_coarse.py:251-286) ----------------------------------------------------------------------------
@_op('gauss_align')
def gauss_align(meta: torch.Tensor, mask: torch.Tensor, gx: Optional[torch.Tensor], tx: float, ratio: float, crops: int, K: int) -> torch.Tensor:
    return _ops._GaussAlign.forward(_Ctx(), meta, mask, gx, tx, ratio, crops, K)


@gauss_align.register_fake
def _(meta, mask, gx, tx, ratio, crops, K):
    return mask.new_empty(mask.shape[0] * crops, mask.shape[1], K, dtype=torch.float32)


@_op('gauss_align_backward')
def gauss_align_backward(gGX: torch.Tensor, meta: torch.Tensor, mask: torch.Tensor, gx: torch.Tensor, tx: float, ratio: float, crops: int,
                         K: int) -> torch.Tensor:
    c = _Ctx((meta.to(torch.int64).contiguous(), mask.contiguous(), gx.contiguous()), cfg=(float(tx), float(ratio), crops, K))
    return _ops._GaussAlign.backward(c, gGX)[2]


@gauss_align_backward.register_fake
def _(gGX, meta, mask, gx, tx, ratio, crops, K):
    return torch.empty_like(gx)


def _ga_setup(ctx, inputs, output):
    meta, mask, gx, tx, ratio, crops, K = inputs
    ctx.save_for_backward(meta, mask, gx)
    ctx.cfg = (tx, ratio, crops, K)


def _ga_backward(ctx, gGX):
    meta, mask, gx = ctx.saved_tensors
    ggx = torch.ops.cfn.gauss_align_backward(gGX, meta, mask, gx, *ctx.cfg) if gx is not None else None
    return None, None, ggx, None, None, None, None


gauss_align.register_autograd(_ga_backward, setup_context=_ga_setup)


# ---- Multi-stage-Fusion gather (x3d_coarse.py:199-247) --------------------------------------------------------------------------------
@_op('fusion_gather')
def fusion_gather(x: torch.Tensor, at_raw: torch.Tensor, at_bias: Optional[torch.Tensor], GX: torch.Tensor, mask: torch.Tensor,
                  crops: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    c = _Ctx()
    z = _ops._FusionGather.forward(c, x, at_raw, at_bias, GX, mask, crops)
    return z, c.saved_tensors[6]


@fusion_gather.register_fake
def _(x, at_raw, at_bias, GX, mask, crops=1):
    B, C, Tf, P = x.shape
    return x.new_empty(B * crops, C, GX.shape[2], P), x.new_empty(B * crops, GX.shape[2], P)


@_op('fusion_gather_backward')
def fusion_gather_backward(gz: torch.Tensor, z: torch.Tensor, den: torch.Tensor, x: torch.Tensor, at_raw: torch.Tensor,
                           at_bias: Optional[torch.Tensor], GX: torch.Tensor, mask: torch.Tensor,
                           crops: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    c = _Ctx((x.contiguous(), at_raw.contiguous(), at_bias, GX.contiguous(), mask.contiguous(), z, den), crops=crops)
    gx, gat, gb, gGX = _ops._FusionGather.backward(c, gz)[:4]
    return gx, gat, _orz(gb, x), gGX


@fusion_gather_backward.register_fake
def _(gz, z, den, x, at_raw, at_bias, GX, mask, crops):
    return torch.empty_like(x), torch.empty_like(at_raw), (torch.empty_like(at_bias) if at_bias is not None else x.new_empty(1)), torch.empty_like(GX)


def _fg_setup(ctx, inputs, output):
    x, at_raw, at_bias, GX, mask, crops = inputs
    ctx.save_for_backward(x, at_raw, at_bias, GX, mask, output[0], output[1])
    ctx.crops = crops


def _fg_backward(ctx, gz, _gden):
    x, at_raw, at_bias, GX, mask, z, den = ctx.saved_tensors
    gx, gat, gb, gGX = torch.ops.cfn.fusion_gather_backward(_gz(gz, z), z, den, x, at_raw, at_bias, GX, mask, ctx.crops)
    return gx, gat, (gb if at_bias is not None else None), gGX, None, None


fusion_gather.register_autograd(_fg_backward, setup_context=_fg_setup)


# ---- block-broadcast FiLM: x * m + c with m, c constant over f x f blocks -----------------------------------------------------------------
@_op('film')
def film(x: torch.Tensor, m: torch.Tensor, c: torch.Tensor, f: int) -> torch.Tensor:
    return _ops._Film.forward(_Ctx(), x, m, c, f)


@film.register_fake
def _(x, m, c, f):
    return torch.empty_like(x)


@_op('film_backward')
def film_backward(g: torch.Tensor, x: torch.Tensor, m: torch.Tensor, f: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return _ops._Film.backward(_Ctx((x.contiguous(), m.contiguous()), f=f), g)[:3]


@film_backward.register_fake
def _(g, x, m, f):
    return torch.empty_like(x), torch.empty_like(m), torch.empty_like(m)


def _film_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.f = inputs[3]


film.register_autograd(lambda ctx, g: (*torch.ops.cfn.film_backward(g, *ctx.saved_tensors, ctx.f), None), setup_context=_film_setup)


# ---- linear resize along t (F.interpolate mode='linear', both align_corners conventions) -------------------------------------------------
@_op('time_resize')
def time_resize(x: torch.Tensor, L: int, align_corners: bool = True) -> torch.Tensor:
    return _ops._TimeResize.forward(_Ctx(), x, L, align_corners)


@time_resize.register_fake
def _(x, L, align_corners=True):
    return x.new_empty(tuple(x.shape[:2]) + (L,) + tuple(x.shape[3:]), dtype=torch.float32)


@_op('time_resize_backward')
def time_resize_backward(g: torch.Tensor, shape: List[int], L: int, align_corners: bool) -> torch.Tensor:
    return _ops._TimeResize.backward(_Ctx(meta=(tuple(shape), L, int(bool(align_corners)))), g)[0]


@time_resize_backward.register_fake
def _(g, shape, L, align_corners):
    return g.new_empty(shape, dtype=torch.float32)


def _tr_setup(ctx, inputs, output):
    ctx.meta = (list(inputs[0].shape), inputs[1], inputs[2])


time_resize.register_autograd(lambda ctx, g: (torch.ops.cfn.time_resize_backward(g, *ctx.meta), None, None), setup_context=_tr_setup)


OPERATORS = ('dwconv3d', 'pwconv', 'time_sample', 'dwconv_t5', 'stem_conv', 'conv3d_dense', 'bn_fold', 'bn_add_relu', 'affine_act',
             'pool_hw', 'interp1d', 'grid_cdf', 'gauss_align', 'fusion_gather', 'film', 'time_resize')


# ---------------------------------------------------------------------------------------------------------------------------
# cfn_hip.ops-shaped facade over the registered operators: the CFN_USE_TORCH_OPS route of the x3d_coarse modules (GridPoolLayer,
# GridUnpool, Gaussian, RewightLayer, MixingLayer; reference x3d_coarse.py:175-451) calls the SAME function names with the same
# arguments, and every call goes through the dispatcher (torch.ops.cfn.*) instead of the autograd Functions of cfn_hip.ops.
# Prologue coefficients arrive as fp64 tensors holding fp32 values (cfn_hip.ops convention); the operators' gradient formulas return
# fp32, so they are narrowed here (lossless, differentiable).
# ---------------------------------------------------------------------------------------------------------------------------
def _f32(t):
    return t if t is None or t.dtype == torch.float32 else t.float()


class TorchOps(object):
    @staticmethod
    def pwconv(x, w, A=None, B=None, act=ACT_NONE, stride=1, stats=True, **_unused):
        y, s, q = torch.ops.cfn.pwconv(x, w, _f32(A), _f32(B), act, stride)
        return (y, s, q) if stats else (y, None, None)

    @staticmethod
    def conv3d_dense(x, w, kernel, stride, padding, A=None, B=None, act=ACT_NONE, stats=True):
        y, s, q = torch.ops.cfn.conv3d_dense(x, w, list(kernel), list(stride), list(padding), _f32(A), _f32(B), act)
        return (y, s, q) if stats else (y, None, None)

    @staticmethod
    def affine_act(x, A, B, act=ACT_NONE):
        return torch.ops.cfn.affine_act(x, _f32(A), _f32(B), act)

    @staticmethod
    def pool_hw(x, OH, OW, A=None, B=None, act=ACT_NONE):
        return torch.ops.cfn.pool_hw(x, OH, OW, _f32(A), _f32(B), act)

    @staticmethod
    def fusion_gather(x, at_raw, at_bias, GX, mask, crops=1):
        return torch.ops.cfn.fusion_gather(x, at_raw, at_bias, GX, mask, crops)[0]

    @staticmethod
    def gauss_align(meta, mask, gx, tx, ratio, crops, K):
        return torch.ops.cfn.gauss_align(meta, mask, gx, 1.0 if tx is None else float(tx), float(ratio), crops, K)

    @staticmethod
    def grid_cdf(g, bias=None):
        return torch.ops.cfn.grid_cdf(g, bias)

    @staticmethod
    def time_sample(x, cdf):
        return torch.ops.cfn.time_sample(x, cdf)

    @staticmethod
    def time_resize(x, L, align_corners=True):
        return torch.ops.cfn.time_resize(x, L, align_corners)

    @staticmethod
    def film(x, m, c, f):
        return torch.ops.cfn.film(x, m, c, f)

    @staticmethod
    def interp1d(x, y, xnew):
        return torch.ops.cfn.interp1d(x, y, xnew)
