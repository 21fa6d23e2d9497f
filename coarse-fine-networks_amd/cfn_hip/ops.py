"""torch.autograd glue over the C ABI: each Function allocates outputs with torch (device memory,
caching allocator, current stream) and hands raw pointers to libcfn_hip.so.  No op here has a CPU
or eager fallback.

Convention shared by the conv ops: the input is read through the per-(n,c) *prologue*
``a = act(A*x + B)`` (A, B: fp32 (N,C) or None) and the op returns the raw conv output ``y``
together with fp64 per-(n,c) ``sum(y)`` / ``sum(y*y)``.  Batch-norm statistics, the SE squeeze and
their gradients are then tiny (N,C) tensor expressions handled by ordinary autograd, while every
full-size tensor is touched only inside the HIP kernels.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import ctypes
import threading

from . import call, call_try, check, query, ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID  # noqa: F401


_scratch_epoch = [0]


class _ZeroArena(object):
    """fp64 zero-filled scratch handed out in slices: the statistics / gradient accumulators of ~400 kernel
    launches per step come out of a few large memsets instead of one tiny fill kernel each.  A chunk stays alive
    as long as any slice of it does (autograd may save them)."""
    CHUNK = 1 << 20

    def __init__(self):
        self.buf, self.off, self.epoch = {}, {}, _scratch_epoch[0]

    def take(self, numel, dev):
        if self.epoch != _scratch_epoch[0]:      # reset_scratch() was called (from any thread): drop the old chunks
            self.buf, self.off, self.epoch = {}, {}, _scratch_epoch[0]
        numel_al = (numel + 1) & ~1
        key = (dev.type, dev.index)
        if key not in self.buf or self.off[key] + numel_al > self.buf[key].numel():
            self.buf[key] = torch.zeros(max(self.CHUNK, numel_al), dtype=torch.float64, device=dev)
            self.off[key] = 0
        o = self.off[key]
        self.off[key] = o + numel_al
        return self.buf[key][o:o + numel]


class _GradCast(object):
    """fp64 weight-gradient accumulators -> fp32 gradients with ONE cast kernel per backward pass.

    The wgrad kernels accumulate into fp64 (zero-filled) buffers.  Casting each of the ~110 buffers on its own costs a
    ~6 us kernel apiece, so the accumulators of one backward pass are carved out of one fp64 chunk that is mirrored
    by an fp32 chunk of the same layout; backward returns views of the fp32 chunk and a callback queued on the
    autograd engine fills it with a single copy when the pass ends.

    A view is only handed out when nobody can read the gradient before that copy (`_gw_buffers`): leaf parameter
    without an existing .grad, no create_graph, no tensor hooks, no post-accumulate hooks other than GradReducer's (which
    flushes before it packs a bucket), and the FIRST gradient of that parameter in the running pass (a parameter used
    twice in one graph would have its two unfilled views summed by autograd mid-pass).  State is per (thread, backward
    pass): a pass that died with an exception cannot leave `scheduled` set, because the next pass is recognised by its
    graph-task id and starts from scratch."""

    def __init__(self):
        self._reset()
        self.last_total = 0

    def _reset(self):
        self.cur, self.live, self.scheduled, self.seen, self.task = {}, [], False, set(), None

    def begin(self):
        """called before every hand-out: a new autograd graph task means a new pass"""
        tid = torch._C._current_graph_task_id()
        if tid != self.task:
            if self.live:      # the previous pass never reached its end-of-backward callback (exception): make what it
                self.flush()   # handed out valid, then forget it
            self._reset()
            self.task = tid

    def first_use(self, w):
        if id(w) in self.seen:
            return False
        self.seen.add(id(w))
        return True

    def take(self, numel, shape, dev):
        al = (numel + 3) & ~3
        key = (dev.type, dev.index)
        ch = self.cur.get(key)
        if ch is None or ch['off'] + al > ch['f64'].numel():
            size = max(int(self.last_total * 1.05) + 1024, al, 1 << 16)
            ch = {'f64': torch.zeros(size, dtype=torch.float64, device=dev),
                  'f32': torch.empty(size, dtype=torch.float32, device=dev), 'off': 0, 'done': 0}
            self.cur[key] = ch
            self.live.append(ch)
        o = ch['off']
        ch['off'] = o + al
        if not self.scheduled:
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            self.scheduled = True
        return ch['f64'][o:o + numel], ch['f32'][o:o + numel].view(shape)

    def flush(self):
        for ch in self.live:
            if ch['off'] > ch['done']:
                ch['f32'][ch['done']:ch['off']].copy_(ch['f64'][ch['done']:ch['off']])
                ch['done'] = ch['off']

    def _end_of_backward(self):
        self.flush()
        self.last_total = max(sum(ch['off'] for ch in self.live), 1)
        # the fp32 chunk now belongs to the gradients that view it; the next pass gets fresh chunks
        self._reset()


class _PerThread(threading.local):
    """scratch state is per calling thread (the reference's DataParallel drives one thread per device; the autograd
    engine runs backward on per-device worker threads) and, inside it, per device"""

    def __init__(self):
        self.arena = _ZeroArena()
        self.gradcast = _GradCast()


_tls = _PerThread()
LAZY_GRAD_CAST = True     # set False to cast every weight gradient immediately (one small kernel each)


def _f64(n, c, dev):
    return _tls.arena.take(n * c, dev).view(n, c)


def _f64pair(n, c, dev):
    """two adjacent (n,c) accumulators, returned together so one cast converts both"""
    t = _tls.arena.take(2 * n * c, dev).view(2, n, c)
    return t, t[0], t[1]


def reset_scratch():
    """forget the zero-filled scratch chunks of EVERY thread (forward runs on the caller's thread, backward on the autograd
    engine's per-device threads): the next accumulator allocates a fresh, zeroed chunk.  hipGraph capture brackets itself
    with this: a chunk zeroed before the capture would not be re-zeroed at replay, and a chunk that lives in the graph's
    memory pool must not serve eager calls afterwards."""
    _scratch_epoch[0] += 1


def flush_grad_casts():
    """make every weight gradient handed out so far in the running backward pass valid (see _GradCast)"""
    _tls.gradcast.flush()


def allow_lazy_grad_cast(params, ok=True):
    """Opt parameters in to the lazily cast weight gradients (_GradCast): their gradient may be handed to autograd as a view
    that is filled at the end of the backward pass.  ONLY for parameters whose every reader inside the pass calls
    flush_grad_casts() first -- cfn_hip.dist.GradReducer does and opts its parameters in.  Anything else (torch DDP / FSDP
    reducers and other hooks on the AccumulateGrad node, which cannot be detected from here) keeps the default: an immediate
    cast, one small kernel per weight gradient."""
    for p in params:
        p._cfn_lazy_grad_ok = bool(ok)


def _lazy_ok(w):
    if not LAZY_GRAD_CAST or w is None or not w.is_leaf or w.grad is not None or torch.is_grad_enabled():
        return False
    if not getattr(w, '_cfn_lazy_grad_ok', False):
        return False      # not opted in (allow_lazy_grad_cast): somebody unseen (DDP / FSDP node hooks) might read it early
    if w._backward_hooks:
        return False
    hooks = getattr(w, '_post_accumulate_grad_hooks', None)
    if hooks:
        from .dist import _BucketHook
        if not all(isinstance(h, _BucketHook) for h in hooks.values()):
            return False  # a foreign post-accumulate hook (optimizer-in-backward ...) would read the gradient before the flush
    return True


def _lazy_ok_probe(w):
    """_lazy_ok as the backward pass would see it (grad mode off); for tests"""
    with torch.no_grad():
        return _lazy_ok(w)


def _grad_owner(w):
    """the leaf parameter whose .grad receives w's gradient without a kernel in between: w itself, or -- for a pure reshape of a whole
    parameter, `conv1d.weight.view(O, I, 1, 1, 1)` of the fusion modules (x3d_coarse._w5), `fc2.weight.view(...)` of the head -- the
    parameter behind the view (ViewBackward only reshapes the gradient).  The lazy-cast rules are the owner's."""
    if w is not None and not w.is_leaf and w._is_view():
        base = w._base
        if base is not None and base.is_leaf and base.numel() == w.numel() and w.is_contiguous() and base.is_contiguous():
            return base
    return w


def _gw_buffers(w, rows, cols, dev):
    """(fp64 accumulator (rows, cols), finish() -> fp32 gradient shaped like w)"""
    gc = _tls.gradcast
    owner = _grad_owner(w)
    if _lazy_ok(owner):
        gc.begin()
        if gc.first_use(owner):
            g64, g32 = gc.take(rows * cols, tuple(w.shape), dev)
            return g64.view(rows, cols), (lambda: g32)
        # second gradient of the same parameter in this pass: autograd is about to ADD it to the view handed out for the
        # first one -- fill that view now (its accumulation was enqueued before this call), and cast this one eagerly
        gc.flush()
    g64 = _f64(rows, cols, dev)
    return g64, (lambda: g64.float().view(tuple(w.shape)))


def _opt(t):
    return None if t is None else t.contiguous()


def _coef(t, rows=None, cols=None):
    """prologue coefficients (N,C): the ABI takes fp64 (bn_fold emits fp64; hand-made fp32 ones are widened here).  rows / cols: the (N, C)
    the kernel will index -- the C ABI sees a pointer only, so a coefficient tensor of another size is refused here instead of being read
    out of bounds on the device"""
    if t is None:
        return None
    if rows is not None and t.numel() != rows * cols:
        raise RuntimeError('per-sample coefficients of %d x %d channels expected, got a tensor of shape %s' % (rows, cols, tuple(t.shape)))
    return t.to(torch.float64).contiguous()


class ShortcutToken(object):
    """Couples the two pointwise convs that read a stage-first block's input (conv1, stride 1, and the spatially
    strided shortcut conv, x3d_fine.py:284-287) in the backward pass: the shortcut conv, whose backward runs first, leaves
    its data gradient as a COMPACT tensor at output resolution here instead of a zero-filled sparse one, and conv1's
    data-gradient kernel adds it on the stride lattice before its act' epilogue (the dense add between the two
    gradients and the memset disappear).  If conv1's backward happens to run first the plain path is used."""

    def __init__(self):
        self.acc, self.acc_stride, self.main_done = None, 1, False


class TailLink(object):
    """Couples a block tail (bn_add_relu) with the pointwise conv(s) whose raw outputs it consumes (conv3 as `y`, the
    strided shortcut conv as `res`).  With a link the tail's backward writes ONE tensor g = (gout [+ gout2]) * relu' and
    hands it out as the gradient of both inputs; the per-(n,c) factors A (for y) and Ar (for res) that turn it into the
    true gradients are left here and applied by the convs' backward kernels when they load it (`gscale`).  The gradient
    tensors of y / res are therefore only meaningful to those convs -- which is why the link must be given to both
    sides."""

    def __init__(self):
        self.y_scale = self.res_scale = None
        self.done = False


def _bf(t):
    """2-byte activation tensor (bf16 or fp16: the *_bf16 / *_f16 entry points)"""
    return t is not None and t.dtype in (torch.bfloat16, torch.float16)


def subsample_hw(x, s):
    """x[..., ::s, ::s] of a bf16 (N,C,T,H,W) tensor (cfn_subsample_hw_bf16); no autograd (used inside the conv ops)"""
    N, C, T, H, W = x.shape
    out = torch.empty(N, C, T, (H - 1) // s + 1, (W - 1) // s + 1, dtype=x.dtype, device=x.device)
    call('cfn_subsample_hw' + _sfx(x), x, out, N * C * T, H, W, s)
    return out


class _PwConv(Function):
    """1x1x1 conv (optionally spatial stride 2); fp32 MFMA for fp32 tensors (cfn_pwconv_*), bf16 MFMA with fp32 accumulation
    for bf16 tensors (cfn_pwconv_*_bf16: a strided conv is a gather + the stride-1 contraction on the compact tensor)."""

    @staticmethod
    def forward(ctx, x, A, B, w, act, stride, want_stats, token, role, tail=None, tail_role=None):
        x = check(x).contiguous()
        N, Cin, T, H, W = x.shape
        Cout = w.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(N, Cout, T, Ho, Wo, dtype=x.dtype, device=x.device)
        s = q = None
        if want_stats:
            s, q = _f64(N, Cout, x.device), _f64(N, Cout, x.device)
        w2 = w.reshape(Cout, Cin).contiguous()
        A, B = _coef(A, N, Cin), _coef(B, N, Cin)
        xs = None
        if _bf(x):
            xs = subsample_hw(x, stride) if stride > 1 else x
            call('cfn_pwconv_fwd' + _sfx(x), xs, A, B, act, w2, y, s, q, N, Cin, Cout, T * Ho * Wo)
        else:
            call('cfn_pwconv_fwd', x, A, B, act, w2, y, s, q, N, Cin, Cout, T, H, W, stride)
        ctx.save_for_backward(x, A, B, w2, y, xs if stride > 1 else None)
        ctx.meta = (act, stride, tuple(w.shape))
        ctx.wparam = w
        ctx.token, ctx.role = token, role
        ctx.tail, ctx.tail_role = tail, tail_role
        if not want_stats:
            return y, None, None
        return y, s, q

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gs, gq):
        x, A, B, w2, y, xs = ctx.saved_tensors
        act, stride, wshape = ctx.meta
        token, role = ctx.token, ctx.role
        N, Cin, T, H, W = x.shape
        Cout = w2.shape[0]
        gy = torch.zeros_like(y) if gy is None else gy.contiguous()
        gs, gq = _opt(gs), _opt(gq)
        gsc = None   # gy of a linked block tail is unscaled: the kernels apply the tail's per-(n,c) factor on load
        if ctx.tail is not None and ctx.tail.done:
            gsc = ctx.tail.y_scale if ctx.tail_role == 'y' else ctx.tail.res_scale
        if _bf(x):
            return _PwConv._backward_bf16(ctx, gy, gs, gq, gsc)
        gx = gA = gB = gw = None
        g64 = fin = None
        fused = False
        if ctx.needs_input_grad[0] or (A is not None and ctx.needs_input_grad[1]):
            if token is not None and role == 'short' and stride > 1 and not token.main_done:
                # compact W^T g' at output resolution (a stride-1 data gradient over the strided grid, no epilogue);
                # conv1's data gradient consumes it
                Ho, Wo = y.shape[3], y.shape[4]
                da = torch.empty(N, Cin, T, Ho, Wo, dtype=torch.float32, device=x.device)
                call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w2, None, None, None, ACT_NONE, da, None, None, N, Cin, Cout, T,
                     Ho, Wo, 1, None, 1, gsc)
                token.acc, token.acc_stride = da, stride
            else:
                gx = torch.zeros_like(x) if stride != 1 else torch.empty_like(x)
                ab = a64 = b64 = None
                if A is not None:
                    ab, a64, b64 = _f64pair(N, Cin, x.device)
                acc, acc_stride = None, 1
                if token is not None and role == 'main':
                    if token.acc is not None:
                        acc, acc_stride, token.acc = token.acc, token.acc_stride, None
                    else:
                        token.main_done = True
                # few channels (layer 1: HBM bound): data and weight gradient in one pass over gy, y, x
                if stride == 1 and ctx.needs_input_grad[3]:
                    g64, fin = _gw_buffers(ctx.wparam, Cout, Cin, x.device)
                    fused = call_try('cfn_pwconv_bwd_fused', gy, y, gs, gq, w2, x, A, B, act, gx, a64, b64, g64, N, Cin, Cout,
                                     T, H, W, acc, acc_stride, gsc)
                if not fused:
                    call('cfn_pwconv_bwd_data_acc', gy, y, gs, gq, w2, x, A, B, act, gx, a64, b64, N, Cin, Cout, T, H, W, stride,
                         acc, acc_stride, gsc)
                if A is not None:
                    gA, gB = ab[0], ab[1]
        if ctx.needs_input_grad[3]:
            if g64 is None:
                g64, fin = _gw_buffers(ctx.wparam, Cout, Cin, x.device)
            if not fused:
                call('cfn_pwconv_bwd_weight', gy, y, gs, gq, x, A, B, act, g64, N, Cin, Cout, T, H, W, stride, gsc)
            gw = fin()
        return gx, gA, gB, gw, None, None, None, None, None, None, None

    @staticmethod
    def _backward_bf16(ctx, gy, gs, gq, gsc):
        """bf16 tensors: data gradient and weight gradient as two bf16-MFMA kernels; a strided conv works on the compact
        (subsampled) input saved by the forward"""
        x, A, B, w2, y, xs = ctx.saved_tensors
        act, stride, wshape = ctx.meta
        token, role = ctx.token, ctx.role
        N, Cin, T, H, W = x.shape
        Cout = w2.shape[0]
        Ho, Wo = y.shape[3], y.shape[4]
        xin = xs if stride > 1 else x
        gx = gA = gB = gw = None
        if ctx.needs_input_grad[0] or (A is not None and ctx.needs_input_grad[1]):
            if token is not None and role == 'short' and stride > 1 and not token.main_done:
                da = torch.empty(N, Cin, T, Ho, Wo, dtype=x.dtype, device=x.device)      # compact, no epilogue
                call('cfn_pwconv_bwd_data' + _sfx(gy), gy, y, gs, gq, w2, None, None, None, ACT_NONE, da, None, None, N, Cin, Cout, T,
                     Ho, Wo, None, 1, gsc)
                token.acc, token.acc_stride = da, stride
            else:
                ab = a64 = b64 = None
                if A is not None:
                    ab, a64, b64 = _f64pair(N, Cin, x.device)
                acc, acc_stride = None, 1
                if token is not None and role == 'main':
                    if token.acc is not None:
                        acc, acc_stride, token.acc = token.acc, token.acc_stride, None
                    else:
                        token.main_done = True
                gxc = torch.empty_like(xin)
                call('cfn_pwconv_bwd_data' + _sfx(gy), gy, y, gs, gq, w2, xin, A, B, act, gxc, a64, b64, N, Cin, Cout, T, Ho, Wo,
                     acc, acc_stride, gsc)
                if stride > 1:      # only reached when conv1's backward ran before the shortcut's: scatter onto the lattice
                    gx = torch.zeros_like(x)
                    gx[:, :, :, ::stride, ::stride] = gxc
                else:
                    gx = gxc
                if A is not None:
                    gA, gB = ab[0], ab[1]
        if ctx.needs_input_grad[3]:
            g64, fin = _gw_buffers(ctx.wparam, Cout, Cin, x.device)
            call('cfn_pwconv_bwd_weight' + _sfx(gy), gy, y, gs, gq, xin, A, B, act, g64, N, Cin, Cout, T * Ho * Wo, gsc)
            gw = fin()
        return gx, gA, gB, gw, None, None, None, None, None, None, None


def pwconv(x, w, A=None, B=None, act=ACT_NONE, stride=1, stats=True, token=None, role=None, tail=None, tail_role=None):
    """returns (y, sum, sumsq); sum/sumsq are None when stats=False.  token / role ('main' | 'short'): see ShortcutToken;
    tail / tail_role ('y' | 'res'): see TailLink.

    The kernels address one sample's (channels x positions) block through a 32-bit buffer descriptor (2 GiB fp32, 1 GiB
    bf16).  Whole-video validation exceeds that (x3d_coarse layer 1 on a 1000-frame chunk, train_coarse_fineFEAT.py:215-224:
    54 x 1000 x 112 x 112 x 4 B = 2.7 GB): without autograd the conv then runs over frame ranges -- a 1x1x1 conv is
    per-frame -- and the statistics add up.  With autograd the kernel's refusal stands."""
    N, Cin, T, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    es = x.element_size()
    lim = (1 << 31) if es == 4 else (1 << 30)
    span = max(Cin * H * W, w.shape[0] * H * W, w.shape[0] * Ho * Wo, Cin * Ho * Wo) * es     # pw_plan range-checks M * Pin too
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, A, B))
    if span * T < lim or needs_grad or T == 1:
        return _PwConv.apply(x, A, B, w, act, stride, stats, token, role, tail, tail_role)
    step = max((lim - 1) // span, 1)
    odd_plane = es == 2 and (Ho * Wo) % 2 == 1       # bf16 kernels need an even position count: even frame counts then
    if odd_plane:
        if 2 * span >= lim:
            raise RuntimeError('pwconv: two frames of %d x %d x %d bf16 elements exceed the 1 GiB buffer range the frame-range '
                               'chunking has to respect (odd plane: frames travel in pairs)' % (max(Cin, w.shape[0]), H, W))
        step = max(step & ~1, 2)
    y = torch.empty(N, w.shape[0], T, Ho, Wo, dtype=x.dtype, device=x.device)
    s = q = None
    t0 = 0
    while t0 < T:
        t1 = min(t0 + step, T)
        if odd_plane and (t1 - t0) % 2 and t1 - t0 > 1:
            t1 -= 1                               # keep the chunk even; a single odd frame is left for the end
        pad = odd_plane and (t1 - t0) % 2 == 1     # T itself odd: the last frame travels with a pad frame (its output, act(B), is sliced off below and the statistics are recomputed)
        xc = x[:, :, t0:t1].contiguous()
        if pad:
            xc = torch.cat([xc, torch.zeros_like(xc)], 2)
        yc, sc, qc = _PwConv.apply(xc, A, B, w, act, stride, False if pad else stats, None, None)
        if pad:
            yc = yc[:, :, :1]
            if stats:
                sc, qc = yc.double().sum((2, 3, 4)), (yc.double() ** 2).sum((2, 3, 4))
        y[:, :, t0:t1] = yc
        if stats:
            s, q = (sc, qc) if s is None else (s + sc, q + qc)
        t0 = t1
    return y, s, q


def _sfx(t):
    """entry-point suffix for the element type of an activation tensor"""
    return '_bf16' if t.dtype == torch.bfloat16 else ('_f16' if t.dtype == torch.float16 else '')


class _DwConv3d(Function):
    """depthwise 3x3x3, pad 1, stride (1,s,s); see cfn_dwconv3d_*."""

    @staticmethod
    def forward(ctx, x, A, B, w, act, stride, want_stats):
        x = check(x).contiguous()
        N, C, T, H, W = x.shape
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        y = torch.empty(N, C, T, Ho, Wo, dtype=x.dtype, device=x.device)
        s = q = None
        if want_stats:
            s, q = _f64(N, C, x.device), _f64(N, C, x.device)
        w2 = w.reshape(C, 27).contiguous()
        A, B = _coef(A, N, C), _coef(B, N, C)
        call('cfn_dwconv3d_fwd' + _sfx(x), x, A, B, act, w2, y, s, q, N, C, T, H, W, stride)
        ctx.save_for_backward(x, A, B, w2, y)
        ctx.meta = (act, stride, tuple(w.shape))
        ctx.wparam = w
        if not want_stats:
            return y, None, None
        return y, s, q

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gs, gq):
        x, A, B, w2, y = ctx.saved_tensors
        act, stride, wshape = ctx.meta
        N, C, T, H, W = x.shape
        gy = torch.zeros_like(y) if gy is None else gy.contiguous()
        gs, gq = _opt(gs), _opt(gq)
        gx = gA = gB = gw = None
        want_x = ctx.needs_input_grad[0] or (A is not None and ctx.needs_input_grad[1])
        want_w = ctx.needs_input_grad[3]
        ab = a64 = b64 = g64 = fin = None
        if want_x:
            gx = torch.empty_like(x)
            if A is not None:
                ab, a64, b64 = _f64pair(N, C, x.device)
        if want_w:
            g64, fin = _gw_buffers(ctx.wparam, C, 27, x.device)
        # stride 1, big planes: data and weight gradient in ONE pass over gy, y, x (declines small planes)
        sfx = _sfx(x)
        fused = want_x and want_w and stride == 1 and call_try('cfn_dwconv3d_bwd_fused' + sfx, gy, y, gs, gq, w2, x, A, B, act,
                                                             gx, a64, b64, g64, N, C, T, H, W)
        # stride 2 (first block of a stage): x read once instead of twice (declines other geometries)
        if not fused and want_x and want_w and stride == 2:
            fused = call_try('cfn_dwconv3d_bwd_fused_s2' + sfx, gy, y, gs, gq, w2, x, A, B, act, gx, a64, b64, g64, N, C, T, H, W)
        if not fused:
            if want_x:
                call('cfn_dwconv3d_bwd_data' + sfx, gy, y, gs, gq, w2, x, A, B, act, gx, a64, b64, N, C, T, H, W, stride)
            if want_w:
                call('cfn_dwconv3d_bwd_weight' + sfx, gy, y, gs, gq, x, A, B, act, g64, N, C, T, H, W, stride)
        if want_x and A is not None:
            gA, gB = ab[0], ab[1]
        if want_w:
            gw = fin()
        return gx, gA, gB, gw, None, None, None


def dwconv3d(x, w, A=None, B=None, act=ACT_NONE, stride=1, stats=True):
    return _DwConv3d.apply(x, A, B, w, act, stride, stats)


class _DwConvT5(Function):
    """depthwise 5x1x1 temporal conv of the stem; see cfn_dwconv_t5_*."""

    @staticmethod
    def forward(ctx, x, w, want_stats, out_dtype=None):
        x = check(x, torch.float32).contiguous()       # the stem conv's output stays fp32 in both precisions
        N, C, T, H, W = x.shape
        y = torch.empty_like(x, dtype=out_dtype or x.dtype)
        s = q = None
        if want_stats:
            s, q = _f64(N, C, x.device), _f64(N, C, x.device)
        w2 = w.reshape(C, 5).contiguous()
        call('cfn_dwconv_t5_fwd' + _sfx(y), x, w2, y, s, q, N, C, T, H * W)
        ctx.save_for_backward(x, w2, y)
        ctx.wshape = tuple(w.shape)
        ctx.wparam = w
        if not want_stats:
            return y, None, None
        return y, s, q

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gs, gq):
        x, w2, y = ctx.saved_tensors
        N, C, T, H, W = x.shape
        gy = torch.zeros_like(y) if gy is None else gy.contiguous()
        gs, gq = _opt(gs), _opt(gq)
        gx = gw = None
        want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if want_x:
            gx = torch.empty_like(x)
        if want_w:
            g64, fin = _gw_buffers(ctx.wparam, C, 5, x.device)
        # both gradients in one march over gy, y, x (declines planes that are not whole float4s)
        fused = want_x and want_w and call_try('cfn_dwconv_t5_bwd_fused' + _sfx(y), gy, y, gs, gq, w2, x, gx, g64, N, C, T, H * W)
        if not fused:
            if want_x:
                call('cfn_dwconv_t5_bwd_data' + _sfx(y), gy, y, gs, gq, w2, gx, N, C, T, H * W)
            if want_w:
                call('cfn_dwconv_t5_bwd_weight' + _sfx(y), gy, y, gs, gq, x, g64, N, C, T, H * W)
        if want_w:
            gw = fin()
        return gx, gw, None, None


def dwconv_t5(x, w, stats=True, out_dtype=None):
    """out_dtype=torch.bfloat16: the bf16 activation path starts here (fp32 in, bf16 out)"""
    return _DwConvT5.apply(x, w, stats, out_dtype)


class _StemConv(Function):
    """conv1_s: dense 1x3x3 stride (1,2,2) pad (0,1,1); the clip gets no gradient."""

    @staticmethod
    def forward(ctx, x, w):
        x = check(x).contiguous()
        N, Ci, T, H, W = x.shape
        Co = w.shape[0]
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty(N, Co, T, Ho, Wo, dtype=torch.float32, device=x.device)
        w2 = w.reshape(Co, Ci * 9).contiguous()
        call('cfn_stem_conv_fwd', x, w2, y, N, Ci, Co, T, H, W)
        ctx.save_for_backward(x)
        ctx.wshape = tuple(w.shape)
        ctx.wparam = w
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        N, Ci, T, H, W = x.shape
        Co = ctx.wshape[0]
        gw = None
        if ctx.needs_input_grad[1]:
            g64, fin = _gw_buffers(ctx.wparam, Co, Ci * 9, x.device)
            call('cfn_stem_conv_bwd_weight', gy.contiguous(), x, g64, N, Ci, Co, T, H, W)
            gw = fin()
        return None, gw


def stem_conv(x, w):
    return _StemConv.apply(x, w)


class _BnFold(Function):
    """(sum, sumsq) of a conv output -> per-(n,c) prologue (A, B); optional fused SE gate.  cfn_bn_fold_*."""

    @staticmethod
    def forward(ctx, s, q, gamma, beta, w1, b1, w2, b2, bufs, cfg):
        training, N, C, S, count, eps, momentum, pool_count = cfg
        run_mean, run_var, nbt = bufs
        dev = run_mean.device
        Wd = w1.shape[0] if w1 is not None else 0
        Se = S if training else 1
        A = torch.empty(N, C, dtype=torch.float64, device=dev)   # prologue coefficients travel as fp64 (fp32 values):
        B = torch.empty_like(A)                                    # their gradients are the kernels' fp64 accumulators, uncast
        mean = torch.empty(Se, C, dtype=torch.float64, device=dev)
        rstd = torch.empty_like(mean)
        A0 = B0 = gate = hbuf = pooled = None
        w1c = w2c = None
        if Wd:
            A0, B0, gate, pooled = (torch.empty(N, C, dtype=torch.float32, device=dev) for _ in range(4))
            hbuf = torch.empty(N, Wd, dtype=torch.float32, device=dev)
            w1c, w2c = w1.reshape(Wd, C).contiguous(), w2.reshape(C, Wd).contiguous()
        s, q = _opt(s), _opt(q)
        call('cfn_bn_fold_fwd', s, q, gamma, beta, run_mean, run_var, nbt if training else None, int(training), N, C, S,
             float(count), float(eps), float(momentum), w1c, b1, w2c, b2, Wd, float(pool_count), A, B, mean, rstd,
             A0, B0, gate, hbuf, pooled)
        ctx.save_for_backward(s, gamma, mean, rstd, A0, B0, gate, hbuf, pooled, w1c, w2c)
        ctx.cfg = (training, N, C, S, Wd, count, pool_count, None if w1 is None else tuple(w1.shape),
                   None if w2 is None else tuple(w2.shape))
        return A, B

    @staticmethod
    @once_differentiable
    def backward(ctx, gA, gB):
        s, gamma, mean, rstd, A0, B0, gate, hbuf, pooled, w1c, w2c = ctx.saved_tensors
        training, N, C, S, Wd, count, pool_count, w1s, w2s = ctx.cfg
        dev = mean.device
        gA = torch.zeros(N, C, dtype=torch.float64, device=dev) if gA is None else gA.contiguous()
        gB = torch.zeros(N, C, dtype=torch.float64, device=dev) if gB is None else gB.contiguous()
        gs = gq = None
        if training:
            gs = torch.empty(N, C, dtype=torch.float64, device=dev)
            gq = torch.empty_like(gs)
        elif Wd:     # eval mode: the statistics are constants, but the SE gate still depends on sum(y) (x3d_fine.py:158)
            gs = torch.empty(N, C, dtype=torch.float64, device=dev)
        gg = gbt = None
        if gamma is not None:
            gg, gbt = torch.empty(C, device=dev), torch.empty(C, device=dev)
        gw1 = gb1 = gw2 = gb2 = tA = tB = None
        if Wd:
            gw1, gb1 = torch.empty(Wd, C, device=dev), torch.empty(Wd, device=dev)
            gw2, gb2 = torch.empty(C, Wd, device=dev), torch.empty(C, device=dev)
            tA, tB = (torch.empty(N, C, dtype=torch.float64, device=dev) for _ in range(2))
        call('cfn_bn_fold_bwd', gA, gB, s, gamma, mean, rstd, A0, B0, gate, hbuf, pooled, w1c, w2c, int(training), N, C, S,
             Wd, float(count), float(pool_count), gs, gq, gg, gbt, gw1, gb1, gw2, gb2, tA, tB)
        if Wd:
            gw1, gw2 = gw1.view(w1s), gw2.view(w2s)
        return gs, gq, gg, gbt, gw1, gb1, gw2, gb2, None, None


def bn_fold(s, q, gamma, beta, bufs, training, N, C, S, count, eps, momentum, se=None, pool_count=1.0):
    """bufs = (running_mean, running_var, num_batches_tracked) of split_bn (training) or bn (eval);
    se = (fc1.weight, fc1.bias, fc2.weight, fc2.bias) or None."""
    w1, b1, w2, b2 = se if se is not None else (None, None, None, None)
    return _BnFold.apply(s, q, gamma, beta, w1, b1, w2, b2, bufs, (training, N, C, S, count, eps, momentum, pool_count))


class _BnAddRelu(Function):
    """out = relu(A*y + B + (Ar*res + Br)); Ar/Br None = plain residual.

    split=True returns the output twice (two tensor objects over one storage): the caller hands one to the next
    block's conv1 and the other to its residual input, so autograd delivers their gradients separately and the backward
    kernel sums them on the fly instead of a 3-pass add kernel in between.  Nothing may write to either in place."""

    @staticmethod
    def forward(ctx, y, A, B, res, Ar, Br, split, link):
        y, res = check(y).contiguous(), check(res).contiguous()
        N, C = y.shape[:2]
        vol = y[0, 0].numel()
        if tuple(res.shape) != tuple(y.shape):
            raise RuntimeError('bn_add_relu: residual %s does not match y %s' % (tuple(res.shape), tuple(y.shape)))
        out = torch.empty_like(y)
        A, B, Ar, Br = _coef(A, N, C), _coef(B, N, C), _coef(Ar, N, C), _coef(Br, N, C)
        mask = None
        sfx = _sfx(y)
        if link is not None:   # ReLU bit mask for the one-tensor backward (1/32 of a tensor instead of re-reading out)
            words = query('cfn_bn_add_relu_mask_words' + sfx, N * C, vol)
            if words > 0:
                mask = torch.empty(words, dtype=torch.int32, device=y.device)
        call('cfn_bn_add_relu_fwd' + sfx, y, A, B, res, Ar, Br, out, mask, N * C, vol)
        ctx.link, ctx.has_mask = link, mask is not None
        ctx.save_for_backward(y, A, res, Ar, mask if mask is not None else out)
        if not split:
            return out
        alias = torch.empty(0, dtype=out.dtype, device=out.device).set_(out.untyped_storage(), out.storage_offset(),
                                                                         out.shape, out.stride())
        return out, alias

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, gout2=None):
        y, A, res, Ar, out = ctx.saved_tensors
        N, C = y.shape[:2]
        vol = y[0, 0].numel()
        if gout is None:
            gout, gout2 = gout2, None
        if gout is None:
            gout = torch.zeros_like(y)
        gout2 = None if gout2 is None else gout2.contiguous()
        t3 = _tls.arena.take(3 * N * C, y.device).view(3, N, C)
        gA, gB = t3[0], t3[1]
        gAr = t3[2] if Ar is not None else None
        gBr = gB if Ar is not None else None
        link = ctx.link
        if link is not None:
            g = torch.empty_like(y)
            mask = out if ctx.has_mask else None
            call('cfn_bn_add_relu_bwd_g' + _sfx(y), gout.contiguous(), gout2, None if ctx.has_mask else out, mask, y,
                 res if Ar is not None else None, g, gA, gB, gAr, N * C, vol)
            # DETACHED: A / Ar come back from saved_tensors with their grad_fn, and that graph contains the conv whose ctx holds this link -- a
            # reference cycle through C++ autograd nodes that Python's collector cannot break (it leaked the ctx objects, the link and the (N, C)
            # factors of every block: ~190 KB per step at 8 clips)
            link.y_scale, link.res_scale, link.done = A.detach(), (None if Ar is None else Ar.detach()), True
            # one tensor, two consumers: a second tensor object over the same storage keeps autograd from accumulating
            # into it in place
            g2 = torch.empty(0, dtype=g.dtype, device=g.device).set_(g.untyped_storage(), g.storage_offset(), g.shape, g.stride())
            return g, gA, gB, g2, gAr, gBr, None, None
        if _bf(y):   # no link (a tail used on its own): one-tensor kernel, then the two per-(n,c) scalings as tensor ops
            g = torch.empty_like(y)
            call('cfn_bn_add_relu_bwd_g' + _sfx(y), gout.contiguous(), gout2, out, None, y, res if Ar is not None else None, g, gA, gB,
                 gAr, N * C, vol)
            shp = (N, C) + (1,) * (y.dim() - 2)
            gres = g if Ar is None else (g.float() * Ar.float().view(shp)).to(y.dtype)
            return (g.float() * A.float().view(shp)).to(y.dtype), gA, gB, gres, gAr, gBr, None, None
        gy, gres = torch.empty_like(y), torch.empty_like(res)
        call('cfn_bn_add_relu_bwd', gout.contiguous(), gout2, out, y, A, res, Ar, gy, gres, gA, gB, gAr, N * C, vol)
        return gy, gA, gB, gres, gAr, gBr, None, None


def bn_add_relu(y, A, B, res, Ar=None, Br=None, split=False, link=None):
    """link (TailLink): one-tensor backward, see TailLink -- the producers of y (and of res when Ar is given) MUST be pwconv
    calls carrying the same link"""
    return _BnAddRelu.apply(y, A, B, res, Ar, Br, split, link)


class _AffineAct(Function):
    @staticmethod
    def forward(ctx, x, A, B, act):
        x = check(x).contiguous()
        N, C = x.shape[:2]
        out = torch.empty_like(x)
        A, B = _coef(A, N, C), _coef(B, N, C)
        call('cfn_affine_act_fwd', x, A, B, act, out, N * C, x[0, 0].numel())
        ctx.save_for_backward(x, A, B)
        ctx.act = act
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, A, B = ctx.saved_tensors
        N, C = x.shape[:2]
        gx = torch.empty_like(x)
        a64, b64 = _f64(N, C, x.device), _f64(N, C, x.device)
        call('cfn_affine_act_bwd', gout.contiguous(), x, A, B, ctx.act, gx, a64, b64, N * C, x[0, 0].numel())
        return gx, a64, b64, None


def affine_act(x, A, B, act=ACT_NONE):
    return _AffineAct.apply(x, A, B, act)


class _ChannelStats(Function):
    """per-(n,c) sum / sumsq (fp64) of a tensor; backward = broadcast (gs + 2 x gq)."""

    @staticmethod
    def forward(ctx, x):
        x = check(x).contiguous()
        N, C = x.shape[:2]
        s, q = _f64(N, C, x.device), _f64(N, C, x.device)
        call('cfn_channel_stats', x, s, q, N * C, x[0, 0].numel())
        ctx.save_for_backward(x)
        return s, q

    @staticmethod
    @once_differentiable
    def backward(ctx, gs, gq):
        (x,) = ctx.saved_tensors
        N, C = x.shape[:2]
        gx = torch.zeros_like(x)
        shp = (N, C) + (1,) * (x.dim() - 2)
        if gs is not None:
            gx = gx + gs.float().view(shp)
        if gq is not None:
            gx = gx + 2.0 * gq.float().view(shp) * x
        return gx


def channel_stats(x):
    return _ChannelStats.apply(x)


class _PoolHW(Function):
    """adaptive (OH,OW) spatial mean of act(A*x+B); see cfn_pool_hw_*."""

    @staticmethod
    def forward(ctx, x, A, B, act, OH, OW):
        x = check(x).contiguous()
        N, C, T, H, W = x.shape
        out = torch.empty(N, C, T, OH, OW, dtype=torch.float32, device=x.device)     # pooled tensors are fp32 in both precisions
        A, B = _coef(A, N, C), _coef(B, N, C)
        call('cfn_pool_hw_fwd' + _sfx(x), x, A, B, act, out, N * C, T, H, W, OH, OW)
        ctx.save_for_backward(x, A, B)
        ctx.meta = (act, OH, OW)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, A, B = ctx.saved_tensors
        act, OH, OW = ctx.meta
        N, C, T, H, W = x.shape
        gx = torch.empty_like(x)
        a64 = b64 = None
        if A is not None:
            a64, b64 = _f64(N, C, x.device), _f64(N, C, x.device)
        call('cfn_pool_hw_bwd' + _sfx(x), gout.contiguous(), x, A, B, act, gx, a64, b64, N * C, T, H, W, OH, OW)
        if A is None:
            return gx, None, None, None, None, None
        return gx, a64, b64, None, None, None


def pool_hw(x, OH, OW, A=None, B=None, act=ACT_NONE):
    return _PoolHW.apply(x, A, B, act, OH, OW)


class _Film(Function):
    """out = x * m + c with m, c constant over f x f spatial blocks ((N,C,T,H/f,W/f))."""

    @staticmethod
    def forward(ctx, x, m, c, f):
        x, m, c = check(x).contiguous(), check(m).contiguous(), check(c).contiguous()
        N, C, T, H, W = x.shape
        want = (N, C, T, H // f if f > 0 else -1, W // f if f > 0 else -1)
        if f <= 0 or H % f or W % f or tuple(m.shape) != want or tuple(c.shape) != want:
            # (the C ABI sees pointers only: block-constant coefficients of another plane size would be read out of bounds -- 96 x 96 clips
            # against 7 x 7 fine features faulted the device before this check)
            raise RuntimeError('film: coefficients %s / %s do not tile x %s with %d x %d blocks (expected %s)'
                               % (tuple(m.shape), tuple(c.shape), tuple(x.shape), f, f, want))
        out = torch.empty_like(x)
        call('cfn_film_fwd', x, m, c, out, N * C, T, H, W, f)
        ctx.save_for_backward(x, m)
        ctx.f = f
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, m = ctx.saved_tensors
        N, C, T, H, W = x.shape
        gx, gm, gc = torch.empty_like(x), torch.empty_like(m), torch.empty_like(m)
        call('cfn_film_bwd', g.contiguous(), x, m, gx, gm, gc, N * C, T, H, W, ctx.f)
        return gx, gm, gc, None


def film(x, m, c, f):
    return _Film.apply(x, m, c, f)


class _TimeSample(Function):
    """Grid Pool / Unpool resampler: x (B,C,Tin,*) , cdf (B,K) -> (B,C,K,*) ; see cfn_time_sample_*."""

    @staticmethod
    def forward(ctx, x, cdf):
        x, cdf = check(x).contiguous(), check(cdf).contiguous()
        B, C, Tin = x.shape[:3]
        K = cdf.shape[1]
        P = x[0, 0, 0].numel()
        out = torch.empty((B, C, K) + tuple(x.shape[3:]), dtype=torch.float32, device=x.device)
        call('cfn_time_sample_fwd', x, cdf, out, B, C, Tin, K, P)
        ctx.save_for_backward(x, cdf)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, cdf = ctx.saved_tensors
        B, C, Tin = x.shape[:3]
        K = cdf.shape[1]
        P = x[0, 0, 0].numel()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g64 = torch.zeros(B, K, dtype=torch.float64, device=x.device) if ctx.needs_input_grad[1] else None
        call('cfn_time_sample_bwd', g.contiguous(), x, cdf, gx, g64, B, C, Tin, K, P)
        return gx, (g64.float() if g64 is not None else None)


def time_sample(x, cdf):
    return _TimeSample.apply(x, cdf)


class _TimePool(Function):
    """t_pool 'avg' / 'max': (B,C,T,H,W) -> (B,C,T//R,H,W)   (cfn_time_pool_*)"""

    @staticmethod
    def forward(ctx, x, mode, R):
        x = check(x, torch.float32).contiguous()
        B, C, T = x.shape[:3]
        P = x[0, 0, 0].numel()
        out = torch.empty((B, C, T // R) + tuple(x.shape[3:]), dtype=torch.float32, device=x.device)
        call('cfn_time_pool_fwd', x, out, mode, B * C, T, R, P)
        ctx.save_for_backward(x)
        ctx.meta = (mode, R)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        mode, R = ctx.meta
        B, C, T = x.shape[:3]
        gx = torch.empty_like(x)
        call('cfn_time_pool_bwd', g.contiguous(), x, gx, mode, B * C, T, R, x[0, 0, 0].numel())
        return gx, None, None


def time_pool(x, mode, R=4):
    """mode 'avg' | 'max'"""
    return _TimePool.apply(x, {'avg': 0, 'max': 1}[mode], R)


def grid_time_index(cdf, Tin):
    """(i0 int32, w1 fp32) the resampler derives from a CDF (bit-exact ATen index arithmetic)."""
    cdf = check(cdf).contiguous()
    i0 = torch.empty(cdf.shape, dtype=torch.int32, device=cdf.device)
    w1 = torch.empty(cdf.shape, dtype=torch.float32, device=cdf.device)
    call('cfn_grid_time_index', cdf, cdf.numel(), Tin, i0, w1)
    return i0, w1


class _Interp1d(Function):
    @staticmethod
    def forward(ctx, x, y, xnew):
        x, y, xnew = check(x).contiguous(), check(y).contiguous(), check(xnew).contiguous()
        B = max(x.shape[0], y.shape[0], xnew.shape[0])
        N, Pq = x.shape[1], xnew.shape[1]
        rows = (int(x.shape[0] > 1), int(y.shape[0] > 1), int(xnew.shape[0] > 1))
        ynew = torch.empty(B, Pq, dtype=torch.float32, device=x.device)
        ind = torch.empty(B, Pq, dtype=torch.int64, device=x.device)
        call('cfn_interp1d_fwd', x, y, xnew, ynew, ind, B, N, Pq, *rows)
        ctx.save_for_backward(x, y, xnew, ind)
        ctx.rows = rows
        ctx.mark_non_differentiable(ind)
        return ynew, ind

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gind):
        x, y, xnew, ind = ctx.saved_tensors
        B, Pq = ind.shape
        N = x.shape[1]
        gx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.zeros_like(y) if ctx.needs_input_grad[1] else None
        gq = torch.zeros_like(xnew) if ctx.needs_input_grad[2] else None
        call('cfn_interp1d_bwd', g.contiguous(), x, y, xnew, ind, gx, gy, gq, B, N, Pq, *ctx.rows)
        return gx, gy, gq


def interp1d(x, y, xnew):
    """2-D inputs (rows broadcast when a tensor has a single row) -> (ynew, ind)."""
    return _Interp1d.apply(x, y, xnew)


class _TimeResize(Function):
    """linear resize along dim 2 (F.interpolate mode='linear', either align_corners convention)."""

    @staticmethod
    def forward(ctx, x, L, align_corners):
        x = check(x).contiguous()
        BC = x.shape[0] * x.shape[1]
        Kin = x.shape[2]
        P = x[0, 0, 0].numel()
        out = torch.empty(tuple(x.shape[:2]) + (L,) + tuple(x.shape[3:]), dtype=torch.float32, device=x.device)
        call('cfn_time_resize_fwd', x, out, BC, Kin, L, P, int(bool(align_corners)))
        ctx.meta = (tuple(x.shape), L, int(bool(align_corners)))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        shape, L, ac = ctx.meta
        gx = torch.empty(shape, dtype=torch.float32, device=g.device)
        P = 1
        for d in shape[3:]:
            P *= d
        call('cfn_time_resize_bwd', g.contiguous(), gx, shape[0] * shape[1], shape[2], L, P, ac)
        return gx, None, None


def time_resize(x, L, align_corners=True):
    return _TimeResize.apply(x, L, align_corners)


def _geom(kernel, stride, padding):
    return (ctypes.c_int * 9)(*(tuple(kernel) + tuple(stride) + tuple(padding)))


class _ConvDense(Function):
    """dense conv3d (no bias) as implicit GEMM; prologue / stats convention of the other convs.  cfn_conv3d_dense_*."""

    @staticmethod
    def forward(ctx, x, A, B, w, act, kernel, stride, padding, want_stats):
        x = check(x).contiguous()
        N, Ci, T, H, W = x.shape
        Co = w.shape[0]
        g = _geom(kernel, stride, padding)
        To = (T + 2 * padding[0] - kernel[0]) // stride[0] + 1
        Ho = (H + 2 * padding[1] - kernel[1]) // stride[1] + 1
        Wo = (W + 2 * padding[2] - kernel[2]) // stride[2] + 1
        y = torch.empty(N, Co, To, Ho, Wo, dtype=torch.float32, device=x.device)
        s = q = None
        if want_stats:
            s, q = _f64(N, Co, x.device), _f64(N, Co, x.device)
        w2 = w.reshape(Co, -1).contiguous()
        A, B = _coef(A, N, Ci), _coef(B, N, Ci)
        call('cfn_conv3d_dense_fwd', x, A, B, act, w2, y, s, q, N, Ci, Co, T, H, W, g)
        ctx.save_for_backward(x, A, B, w2, y)
        ctx.meta = (act, tuple(kernel), tuple(stride), tuple(padding), tuple(w.shape))
        ctx.wparam = w
        if not want_stats:
            return y, None, None
        return y, s, q

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gs, gq):
        x, A, B, w2, y = ctx.saved_tensors
        act, kernel, stride, padding, wshape = ctx.meta
        N, Ci, T, H, W = x.shape
        Co = w2.shape[0]
        g = _geom(kernel, stride, padding)
        gy = torch.zeros_like(y) if gy is None else gy.contiguous()
        gs, gq = _opt(gs), _opt(gq)
        gx = gA = gB = gw = None
        if ctx.needs_input_grad[0] or (A is not None and ctx.needs_input_grad[1]):
            gx = torch.empty_like(x)
            ab = a64 = b64 = None
            if A is not None:
                ab, a64, b64 = _f64pair(N, Ci, x.device)
            call('cfn_conv3d_dense_bwd_data', gy, y, gs, gq, w2, x, A, B, act, gx, a64, b64, N, Ci, Co, T, H, W, g)
            if A is not None:
                gA, gB = ab[0], ab[1]
        if ctx.needs_input_grad[3]:
            g64, fin = _gw_buffers(ctx.wparam, Co, w2.shape[1], x.device)
            call('cfn_conv3d_dense_bwd_weight', gy, y, gs, gq, x, A, B, act, g64, N, Ci, Co, T, H, W, g)
            gw = fin()
        return gx, gA, gB, gw, None, None, None, None, None


def conv3d_dense(x, w, kernel, stride, padding, A=None, B=None, act=ACT_NONE, stats=True):
    return _ConvDense.apply(x, A, B, w, act, kernel, stride, padding, stats)


class _FusionGather(Function):
    """z[r,c,k,p] = sum_t x[b] at[b] GX[r] mask[b] / (sum_t at GX mask + 1e-6), at = sigmoid(at_raw + at_bias), r = b*crops + j
    (cfn_fusion_gather_*)"""

    @staticmethod
    def forward(ctx, x, at_raw, at_bias, GX, mask, crops):
        x, at_raw, GX, mask = check(x).contiguous(), check(at_raw).contiguous(), check(GX).contiguous(), check(mask).contiguous()
        B, C, Tf, P = x.shape
        K = GX.shape[2]
        if GX.shape[0] != B * crops or GX.shape[1] != Tf or tuple(mask.shape) != (B, Tf) or tuple(at_raw.shape) != (B, Tf, P):
            raise RuntimeError('fusion_gather: inconsistent shapes x %s at %s GX %s mask %s crops %d'
                               % (tuple(x.shape), tuple(at_raw.shape), tuple(GX.shape), tuple(mask.shape), crops))
        z = torch.empty(B * crops, C, K, P, dtype=torch.float32, device=x.device)
        den = torch.empty(B * crops, K, P, dtype=torch.float32, device=x.device)
        call('cfn_fusion_gather_fwd', x, at_raw, at_bias, GX, mask, z, den, B, crops, C, Tf, K, P)
        ctx.save_for_backward(x, at_raw, at_bias, GX, mask, z, den)
        ctx.crops = crops
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, gz):
        x, at_raw, at_bias, GX, mask, z, den = ctx.saved_tensors
        B, C, Tf, P = x.shape
        K, crops = GX.shape[2], ctx.crops
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        want_at = need[1] or (at_bias is not None and need[2])
        gat = torch.empty_like(at_raw) if want_at else None
        gGX = torch.empty_like(GX) if need[3] else None
        dw = torch.empty(B * crops, Tf, K, P, dtype=torch.float32, device=x.device)
        call('cfn_fusion_gather_bwd', gz.contiguous(), z, den, x, at_raw, at_bias, GX, mask, gx, gat, gGX, dw, B, crops, C, Tf, K, P)
        gb = gat.sum().view(at_bias.shape) if (at_bias is not None and need[2]) else None
        return gx, (gat if need[1] else None), gb, gGX, None, None


def fusion_gather(x, at_raw, at_bias, GX, mask, crops=1):
    """x (B,C,Tf,P), at_raw (B,Tf,P), at_bias (1,) or None, GX (B*crops,Tf,K), mask (B,Tf) -> (B*crops,C,K,P)"""
    return _FusionGather.apply(x, at_raw, at_bias, GX, mask, crops)


class _GaussAlign(Function):
    """Gaussian.forward (x3d_coarse.py:256-286) as one kernel; gradient w.r.t. the CDF knots only."""

    @staticmethod
    def forward(ctx, meta, mask, gx, tx, ratio, crops, K):
        meta, mask = meta.contiguous(), check(mask).contiguous()
        if meta.dtype != torch.int64:
            meta = meta.to(torch.int64)
        B, Tf = mask.shape
        if gx is not None:
            gx = check(gx).contiguous()
        GX = torch.empty(B * crops, Tf, K, dtype=torch.float32, device=mask.device)
        call('cfn_gauss_align_fwd', meta, mask, gx, float(tx), float(ratio), GX, B, crops, Tf, K)
        ctx.save_for_backward(meta, mask, gx)
        ctx.cfg = (float(tx), float(ratio), crops, K)
        return GX

    @staticmethod
    @once_differentiable
    def backward(ctx, gGX):
        meta, mask, gx = ctx.saved_tensors
        tx, ratio, crops, K = ctx.cfg
        ggx = None
        if gx is not None and ctx.needs_input_grad[2]:
            B, Tf = mask.shape
            ggx = torch.empty_like(gx)
            call('cfn_gauss_align_bwd', gGX.contiguous(), meta, mask, gx, tx, ratio, ggx, B, crops, Tf, K)
        return None, None, ggx, None, None, None, None


def gauss_align(meta, mask, gx, tx, ratio, crops, K):
    """meta (B,4) int64, mask (B,Tf), gx (B*crops,K) or None (then tl = arange(K)) -> GX (B*crops,Tf,K)"""
    return _GaussAlign.apply(meta, mask, gx, 1.0 if tx is None else tx, ratio, crops, K)


class _GridCdf(Function):
    """saliency logits (B,Kin) (+ scalar bias) -> CDF knots (B,Kin+1)   (cfn_grid_cdf_*)"""

    @staticmethod
    def forward(ctx, g, bias):
        g = check(g).contiguous()
        B, Kin = g.shape
        cdf = torch.empty(B, Kin + 1, dtype=torch.float32, device=g.device)
        call('cfn_grid_cdf_fwd', g, bias, cdf, B, Kin)
        ctx.save_for_backward(g, bias)
        return cdf

    @staticmethod
    @once_differentiable
    def backward(ctx, gcdf):
        g, bias = ctx.saved_tensors
        B, Kin = g.shape
        gg = torch.empty_like(g)
        call('cfn_grid_cdf_bwd', gcdf.contiguous(), g, bias, gg, B, Kin)
        gb = gg.sum().view(bias.shape) if (bias is not None and ctx.needs_input_grad[1]) else None
        return gg, gb


def grid_cdf(g, bias=None):
    return _GridCdf.apply(g, bias)
