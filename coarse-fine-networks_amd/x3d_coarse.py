"""x3d_coarse -- the Coarse stream of Coarse-Fine Networks on MI355X (drop-in for the reference's
``x3d_coarse.py``): X3D trunk + Grid Pool / Grid Unpool + Multi-stage Fusion, same ``generate_model`` /
``ResNet.forward([x, feat, feat_masks, i, meta])`` surface and state_dict keys.

Hot-path design (all full-size tensors go through the gfx950 kernels of ``cfn_hip``):
  * trunk: shared with ``x3d_fine`` (deferred BN/activation prologues, MFMA pointwise, LDS-tiled depthwise);
  * Grid Pool: the saliency convolutions are implicit-GEMM dense convs; the 5-D ``grid_sample`` is a 2-tap
    temporal lerp whose frame indices are bit-identical to ATen's (x3d_coarse.py:394-403);
  * fusion: the reference up-samples the 7x7 fine features to 56/28/14 and materialises a
    (B,C,T',K,h,w) product (1.3 GB / sample for rw2).  All of it is constant over the up-sampling blocks,
    so the branch is evaluated once at 7x7 (gather kernel + MFMA pointwise convs) and applied by a
    block-broadcast FiLM kernel; biases ride in the next op's prologue;
  * Grid Unpool: Interp1d (bit-exact indices) + the same temporal lerp + temporal linear resize.
"""
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from cfn_hip import ops, ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID
from interp1d import Interp1d
from x3d_fine import (SubBatchNorm3d, Swish, Bottleneck, Deferred, conv3x3x3, conv1x1x1, _count,  # noqa: F401
                      get_inplanes, get_blocks)
import x3d_fine

FUSION_HW = 7     # spatial size of the pre-extracted fine features (extract_fineFEAT / x3d_fine.py:345-363)


def _o():
    """operator namespace of the Grid Pool / Unpool / fusion modules: cfn_hip.ops (autograd Functions over the C ABI, the default)
    or, with CFN_USE_TORCH_OPS=1 / x3d_fine.USE_TORCH_OPS = True, the registered dispatcher operators torch.ops.cfn.* behind the
    same function names (cfn_hip.torchlib.TorchOps) -- the same kernels, visible to torch.compile / torch.export"""
    if x3d_fine.USE_TORCH_OPS:
        from cfn_hip import torchlib
        return torchlib.TorchOps
    return ops


def _w5(conv1d):
    """Conv1d (O,I,1) weight as a 1x1x1 conv weight"""
    w = conv1d.weight
    return w.view(w.shape[0], w.shape[1], 1, 1, 1)


class _Rows64(torch.autograd.Function):
    """(C,) fp32 parameter -> (n, C) fp64 prologue coefficient in ONE launch (broadcast + widening copy); the backward is one
    reduction straight to fp32.  (expand + contiguous + the ABI's fp64 conversion and their backwards were five launches.)"""

    @staticmethod
    def forward(ctx, vec, n):
        out = torch.empty(n, vec.numel(), dtype=torch.float64, device=vec.device)
        out.copy_(vec.detach().view(1, -1))
        ctx.dt = vec.dtype
        ctx.shape = tuple(vec.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        return torch.sum(g, 0, dtype=ctx.dt).view(ctx.shape), None


_BANK_INDEX = {}


def _bank_index(n, widths, device):
    """gather index of `_RowsBank` and the blocks' start offsets: output group g is the (n, widths[g]) block at starts[g] (rounded up to 32
    elements = 256 bytes, the alignment a tensor of its own would have), element (r, c) of it reads position sum(widths[:g]) + c of the
    concatenated vectors.  Cached per (n, widths, device) outside graph capture, like `_ones`."""
    key = (n, widths, str(device))
    hit = _BANK_INDEX.get(key)
    if hit is None:
        total = sum(widths)
        grid = torch.arange(total, device=device).view(1, total).expand(n, total)
        parts, starts, o, pos = [], [], 0, 0
        for wd in widths:
            pad = -pos % 32
            if pad:
                parts.append(torch.zeros(pad, dtype=torch.int64, device=device))
            starts.append(pos + pad)
            parts.append(grid[:, o:o + wd].reshape(-1))
            pos += pad + n * wd
            o += wd
        hit = (torch.cat(parts), tuple(starts))
        if not (torch.device(device).type == 'cuda' and torch.cuda.is_current_stream_capturing()):
            _BANK_INDEX[key] = hit
    return hit


class _RowsBank(torch.autograd.Function):
    """`_Rows64` for MANY vectors at once: groups of (C_i,) fp32 vectors -> one (n, sum C_i) fp64 prologue-coefficient block per group (the
    vectors of a group side by side).  Three launches for the whole bank (concatenate, widen, gather) and three in the backward (concatenate
    the blocks' gradients, one column sum in fp64, narrow; the vectors' gradients are views of that one result) -- the fusion branch asks for
    ~25 such blocks per step, three launches each as single `_Rows64` calls."""

    @staticmethod
    def forward(ctx, n, group_sizes, *vecs):
        flat = torch.cat([v.detach().reshape(-1) for v in vecs]).to(torch.float64)
        widths, k = [], 0
        for gs in group_sizes:
            widths.append(sum(v.numel() for v in vecs[k:k + gs]))
            k += gs
        widths = tuple(widths)
        idx, starts = _bank_index(n, widths, flat.device)
        out = torch.index_select(flat, 0, idx)
        ctx.n, ctx.widths = n, widths
        ctx.vec_meta = [(tuple(v.shape), v.dtype) for v in vecs]
        return tuple(out[st:st + n * wd].view(n, wd) for st, wd in zip(starts, widths))

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        gs = [g if g is not None else ref.new_zeros(ctx.n, wd) for g, wd in zip(gs, ctx.widths)]
        total = torch.cat(gs, dim=1).sum(0)
        by_dt = {dt: total.to(dt) for dt in set(dt for _, dt in ctx.vec_meta)}
        grads, o = [], 0
        for shape, dt in ctx.vec_meta:
            c = 1
            for d in shape:
                c *= d
            grads.append(by_dt[dt][o:o + c].view(shape))
            o += c
        return (None, None) + tuple(grads)


_BANK = threading.local()


class _row_bank(object):
    """`with _row_bank(n, groups):` -- every `_rows(vec, n)` / `_rows_of(vecs, n)` call inside finds its block in ONE `_RowsBank` result instead
    of launching its own widening copy.  groups: list of lists of (C_i,) parameters; a request is matched by the identity of its vectors
    and by n, anything else falls back to `_Rows64`."""

    def __init__(self, n, groups):
        seen, self.groups = set(), []
        for g in groups:
            key = tuple(id(v) for v in g)
            if key not in seen and all(v is not None for v in g):
                seen.add(key)
                self.groups.append(list(g))
        self.n, self.blocks = n, None

    def get(self, vecs, n):
        if n != self.n:
            return None
        if self.blocks is None:       # built at the first request: a forward that never asks launches nothing
            flat = [v for g in self.groups for v in g]
            out = _RowsBank.apply(self.n, tuple(len(g) for g in self.groups), *flat)
            self.blocks = {tuple(id(v) for v in g): o for g, o in zip(self.groups, out)}
        return self.blocks.get(tuple(id(v) for v in vecs))

    def __enter__(self):
        self.prev = getattr(_BANK, 'cur', None)
        _BANK.cur = self
        return self

    def __exit__(self, *exc):
        _BANK.cur = self.prev
        return False


def _rows(vec, n):
    """(C,) parameter -> (n, C) per-sample prologue coefficient (fp64 holding fp32 values: what the C ABI takes)"""
    bank = getattr(_BANK, 'cur', None)
    if bank is not None:
        out = bank.get((vec,), n)
        if out is not None:
            return out
    return _Rows64.apply(vec, n)


def _rows_of(vecs, n):
    """`_rows` of the concatenation of several parameters"""
    bank = getattr(_BANK, 'cur', None)
    if bank is not None:
        out = bank.get(tuple(vecs), n)
        if out is not None:
            return out
    return _Rows64.apply(torch.cat(list(vecs)), n)


_ONES = {}


def _ones(n, c, device):
    """constant (n, C) fp64 ones (identity prologue scale), cached per shape and device: no fill / widening launches per use"""
    key = (n, c, str(device))
    t = _ONES.get(key)
    if t is None:
        t = torch.ones(n, c, dtype=torch.float64, device=device)
        if torch.device(device).type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return t      # born in a hipGraph's private pool, filled on replay only: never shared with eager code / other graphs
        _ONES[key] = t
    return t


class RewightLayer(nn.Module):
    """Temporal-alignment gather + two small MLP heads (x3d_coarse.py:175-247)."""

    def __init__(self, channels, g_channels, depth, height, pool=False):
        super(RewightLayer, self).__init__()
        self.at1 = nn.Conv1d(depth, depth, kernel_size=1)
        self.at2 = nn.Conv1d(depth, 1, kernel_size=1)
        self.fc1 = nn.Conv1d(depth, depth, kernel_size=1)
        self.fc2 = nn.Conv1d(depth, channels, kernel_size=1)
        if g_channels is not None:
            self.fc3 = nn.Conv1d(depth, depth, kernel_size=1)
            self.fc4 = nn.Conv1d(depth, g_channels, kernel_size=1)
        self.dropout = nn.Dropout(0.5)
        self.depth, self.height, self.channels, self.g_channels, self.pool = depth, height, channels, g_channels, pool

    def _mlp(self, z5, first, second):
        n = z5.shape[0]
        h, _, _ = _o().pwconv(z5, _w5(first), stats=False)
        one = _ones(n, h.shape[1], z5.device)
        if self.pool and self.training and self.dropout.p > 0:     # x3d_coarse.py:232-233 (rw6 only)
            h = self.dropout(_o().affine_act(h, one, _rows(first.bias, n), ACT_RELU))
            out, _, _ = _o().pwconv(h, _w5(second), stats=False)
        else:
            out, _, _ = _o().pwconv(h, _w5(second), one, _rows(first.bias, n), ACT_RELU, stats=False)
        return out                                                  # raw: the bias of `second` is still to be added

    def gather(self, x, b2, mask, GX):
        """fine features (B,C,T',7,7) -> aligned (b2,C,K,7,7)  (x3d_coarse.py:204-223 at native resolution).  b2 = n*B at
        multi-crop validation (:209-211): the crops of a video share its features / mask / attention and differ in GX."""
        b, c, t, h, w = x.shape
        if mask.shape[1] != t:
            mask = F.adaptive_max_pool1d(mask.unsqueeze(1), t).squeeze(1)
            GX = F.adaptive_avg_pool2d(GX.unsqueeze(1), (t, None)).squeeze(1)
        y1, _, _ = _o().pwconv(x, _w5(self.at1), stats=False)
        one = _ones(b, c, x.device)
        y2, _, _ = _o().pwconv(y1, _w5(self.at2), one, _rows(self.at1.bias, b), ACT_RELU, stats=False)
        # sigmoid(at2 + bias), the mask multiply and the per-crop repeat happen inside the gather kernel
        z = _o().fusion_gather(x.reshape(b, c, t, h * w), y2.view(b, t, h * w), self.at2.bias, GX, mask, b2 // b)
        return z.view(b2, c, GX.shape[2], h, w)

    def forward7(self, x, b2, mask, GX, is_mixing):
        """-> ((bias_raw, bias_b), (scale_raw, scale_b)) at the fine features' own resolution; the Conv1d output
        biases are returned separately so that the consumer folds them into its prologue."""
        z5 = self.gather(x, b2, mask, GX)
        if self.pool:
            z5 = z5.mean(dim=(3, 4), keepdim=True)
        x1 = self._mlp(z5, self.fc1, self.fc2)
        if self.g_channels is None:
            return (x1, self.fc2.bias), None
        x2 = self._mlp(z5, self.fc3, self.fc4)
        return (x1, self.fc2.bias), (x2, self.fc4.bias)

    def forward(self, inp):
        """reference signature: [x_fine, lx, mask, gx, i, GX, isMixing] -> tensors at (height, height)"""
        x, lx, mask, _gx, _i, GX, is_mixing = inp
        (x1, b1), sc = self.forward7(x, lx.shape[0], mask, GX, is_mixing)
        up = (lambda v: v) if self.pool else (lambda v: _upsample(v, self.height))
        x1 = up(x1 + b1.view(1, -1, 1, 1, 1))
        if sc is None:
            return x1
        x2 = sc[0] + sc[1].view(1, -1, 1, 1, 1)
        if not is_mixing:
            x2 = torch.sigmoid(x2)
        return x1, up(x2)


def _upsample(v, height):
    """7x7 -> height x height block replication (= adaptive_max_pool2d up-sampling of the reference)"""
    f = height // v.shape[3]
    if f <= 1:
        return v
    return v.repeat_interleave(f, dim=3).repeat_interleave(f, dim=4)


class Gaussian(nn.Module):
    """Temporal Gaussian alignment weights (x3d_coarse.py:251-286): [meta (B,4), mask (B,T'), gx, tx] -> GX (b2,T',K).
    gx is the CDF (b2,K) with tx the coarse clip length (grid mode), or any tensor whose dim 2 is the coarse length with
    tx=None.  b2 = n*B at multi-crop validation: crop j of a video starts at start + step*j (:264-266).  One HIP kernel
    (cfn_gauss_align_*); the CDF receives its gradient through it."""

    def __init__(self, ratio=1):
        super(Gaussian, self).__init__()
        self.ratio = ratio

    def forward(self, inp):
        meta, mask, gx, tx = inp
        b, b2 = meta.shape[0], gx.shape[0]
        if tx is not None:
            return _o().gauss_align(meta, mask, gx, tx, self.ratio, b2 // b, gx.shape[1])
        return _o().gauss_align(meta, mask, None, None, self.ratio, b2 // b, gx.shape[2])


class MixingLayer(nn.Module):
    """Learned mixing of the four abstraction levels (x3d_coarse.py:289-351, learned path)."""

    def __init__(self, depth, learned=False, index=0, isLogit=False):
        super(MixingLayer, self).__init__()
        self.learned, self.index, self.isLogit = learned, index, isLogit
        self.in_depth = 432 if isLogit else (24 + 48 + 96 + 192)
        self.range = 1 if isLogit else 4
        self.dropout = nn.Dropout(0.5)
        if learned:
            self.conv_at = nn.Conv1d(self.in_depth, depth, kernel_size=1)
            self.conv_at2 = nn.Conv1d(self.in_depth, depth, kernel_size=1)

    @staticmethod
    def prepare(items):
        """list of (raw (B,c_i,K,7,7), bias (c_i,)) -> (concatenated raw tensor, (B, sum c_i) prologue rows of the concatenated biases).  The four
        mixing layers consume the SAME two lists (x3d_coarse.py:700-716): the caller builds this once per list and hands it to every layer
        (`forward7(..., prepared=True)`) -- one cat / one bias-row op per list and step instead of one per layer, and one gradient accumulation
        per layer into the shared tensor instead of one per layer and level."""
        n = items[0][0].shape[0]
        return torch.cat([r for r, _ in items], dim=1), _rows_of([bb for _, bb in items], n)

    def forward7(self, bias, scale, prepared=False):
        """bias / scale: lists of (raw (B,c_i,K,7,7), bias (c_i,)) -- or, with prepared=True, the results of `prepare` on those lists
        -> FiLM coefficients (c7, m7) (B,depth,K,7,7)"""
        if not self.learned:
            raise NotImplementedError('non-learned mixing (x3d_coarse.py:338-344) is unused by the reference scripts')
        if not prepared:
            bias, scale = self.prepare(bias), self.prepare(scale)
        n = bias[0].shape[0]
        one = _ones(n, self.in_depth, bias[0].device)

        def mix(prep, conv, act):
            raw, b = prep
            y, _, _ = _o().pwconv(raw, _w5(conv), one, b, ACT_NONE, stats=False)
            return _o().affine_act(y, _ones(n, y.shape[1], y.device), _rows(conv.bias, n), act)

        return mix(bias, self.conv_at, ACT_NONE), mix(scale, self.conv_at2, ACT_SIGMOID)

    def forward(self, inp):
        """reference signature [x, bias_list, scale_list] with full-resolution block-constant inputs -> (cs, ms) like x"""
        x, bias, scale = inp
        h = x.shape[3]
        zero = lambda v: torch.zeros(v.shape[1], device=v.device)
        down = lambda v: (v[:, :, :, ::v.shape[3] // FUSION_HW, ::v.shape[4] // FUSION_HW].contiguous(), zero(v))
        c7, m7 = self.forward7([down(v) for v in bias], [down(v) for v in scale])
        return _upsample(c7, h), _upsample(m7, h)


class _ZeroGradFor(torch.autograd.Function):
    """identity on `t` that makes `param` a graph input with an exactly-zero gradient (so that the optimizer's weight decay / momentum see the
    parameter as they do in the reference, where its gradient is zero up to rounding)"""

    @staticmethod
    def forward(ctx, t, param):
        ctx.meta = (tuple(param.shape), param.dtype, param.device)
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        shape, dt, dev = ctx.meta
        return g, torch.zeros(shape, dtype=dt, device=dev)


class GridPoolLayer(nn.Module):
    """Learned temporal resampler (x3d_coarse.py:355-416)."""

    def __init__(self, ratio, depth):
        super(GridPoolLayer, self).__init__()
        self.ratio = 4
        self.depth = depth
        self.conv1 = nn.Conv3d(depth, depth, kernel_size=(3, 3, 3), stride=(self.ratio // 2, 2, 2), padding=1)
        self.bn1 = SubBatchNorm3d(num_splits=1, num_features=depth, affine=True)
        self.conv2 = nn.Conv3d(depth, depth, kernel_size=(3, 3, 3), stride=(self.ratio // 2, 2, 2), padding=1)
        self.bn2 = SubBatchNorm3d(num_splits=1, num_features=depth, affine=True)
        self.conv3 = nn.Conv3d(depth, 1, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.relu = nn.ReLU(inplace=True)
        self.sigmoid = nn.Sigmoid()

    def _conv_bn(self, x, conv, bn, A, B, act):
        """conv (+bias) -> BN folded to the next prologue.  The kernel has no bias.  Training: a per-channel constant cancels inside a
        batch-statistics BN -- BN(y + b) = A y + (beta - A mean(y)) for every b -- so the statistics of the bias-free y give the prologue as
        they are; only the running mean sees b (mean(y + b) = mean(y) + b), and the gradient of b is exactly zero.  (Rounds 2-5 shifted the
        sums by b and let autograd differentiate the shift: ~20 small launches per conv for a gradient of rounding noise.)  Eval:
        A (y + b) + B = A y + (B + A b)."""
        n = x.shape[0]
        st = tuple(conv.stride)
        y, s, q = _o().conv3d_dense(x, conv.weight, (3, 3, 3), st, (1, 1, 1), A, B, act, stats=self.training)
        A2, B2 = (bn.fold_op if x3d_fine.USE_TORCH_OPS else bn.fold)(s, q, _count(y), n)
        if not self.training:
            return y, A2, torch.addcmul(B2, A2, conv.bias.to(B2.dtype).view(1, -1))
        with torch.no_grad():
            bn.split_bn.running_mean.view(-1, bn.num_features).add_(conv.bias.view(1, -1), alpha=bn.momentum)
        return y, A2, _ZeroGradFor.apply(B2, conv.bias)

    def saliency(self, x):
        """(B,C,T,H,W) -> (B, T/4) saliency logits (x3d_coarse.py:379-383)"""
        y1, A1, B1 = self._conv_bn(x, self.conv1, self.bn1, None, None, ACT_NONE)
        y2, A2, B2 = self._conv_bn(y1, self.conv2, self.bn2, A1, B1, ACT_RELU)
        y3, _, _ = _o().conv3d_dense(y2, self.conv3.weight, (1, 3, 3), (1, 2, 2), (0, 1, 1), A2, B2, ACT_RELU, stats=False)
        g = _o().pool_hw(y3, 1, 1)
        return g.view(g.shape[0], g.shape[2])          # conv3's bias is added by the CDF kernel

    @staticmethod
    def cdf(g, bias=None):
        """saliency logits (+ bias) -> CDF knots (B, K) (x3d_coarse.py:384-392): one kernel, cumsum accumulated in fp64"""
        return _o().grid_cdf(g, bias)

    def forward(self, inp):
        x = inp.materialize() if isinstance(inp, Deferred) else inp
        gx_out = self.cdf(self.saliency(x), self.conv3.bias)
        return _o().time_sample(x, gx_out), gx_out


def GridUnpool(inp, return_aux=False):
    """inverse temporal grid (x3d_coarse.py:419-451)"""
    x, gx, is_logit = inp
    ratio = 4
    b, k = gx.shape
    mid = torch.arange(k, device=gx.device).to(torch.float32)
    mid = (mid / (k - 1.)).view(1, -1).repeat(b, 1)
    gx_inv, ind = Interp1d().forward(gx, mid, mid, None, return_index=True)
    if is_logit:
        y = _o().time_sample(x.unsqueeze(3), gx_inv).squeeze(3)
    else:
        y = _o().time_resize(_o().time_sample(x, gx_inv), x.shape[2] * ratio)
    if return_aux:
        return y, gx_inv, ind
    return y


class ResNet(x3d_fine.ResNet):
    """Coarse stream (x3d_coarse.py:455-727); the X3D trunk, head and BN bookkeeping are x3d_fine's."""

    def __init__(self, block, layers, block_inplanes, n_input_channels=3, feat_depth={}, conv1_t_size=7,
                 conv1_t_stride=1, shortcut_type='B', widen_factor=1.0, dropout=0.5, n_classes=400, base_bn_splits=8,
                 task='class', extract_feat=False, t_pool=None, learnedMixing=False, isMixing=False, act_dtype=None):
        # act_dtype (an addition to the reference's signature, as in x3d_fine): 'bf16' / 'fp16' = the stem's temporal conv and layer 1 -- the part of
        # the coarse stream that runs at the clip's full frame count, ~3/4 of its bytes -- store their activations and activation gradients as 2-byte
        # elements (16-bit MFMA pointwise products, fp32 accumulation / statistics / weights); the output of layer 1 is widened once, in front of
        # Grid Pool, and everything behind it (saliency convs, resampler, layers 2-4 at T/4 + 1 frames, fusion, head) stays fp32.
        self.feat_depth, self.learnedMixing, self.isMixing, self.t_pool = feat_depth, learnedMixing, isMixing, t_pool
        nn.Module.__init__(self)
        planes = [(int(x * widen_factor), int(y * widen_factor)) for x, y in block_inplanes]
        if t_pool == 'avg':       # parameter-free holders as in the reference (x3d_coarse.py:489-492); the op is cfn_time_pool_*
            self.pool_1 = nn.AvgPool3d((4, 1, 1), stride=(4, 1, 1))
        elif t_pool == 'max':
            self.pool_1 = nn.MaxPool3d((4, 1, 1), stride=(4, 1, 1))
        elif t_pool == 'grid':
            self.pool_1 = GridPoolLayer(ratio=4, depth=planes[0][1])      # registered first, as in the reference
        x3d_fine.ResNet.__init__(self, block, layers, block_inplanes, n_input_channels=n_input_channels,
                                 shortcut_type=shortcut_type, widen_factor=widen_factor, dropout=dropout,
                                 n_classes=n_classes, base_bn_splits=base_bn_splits, task=task,
                                 extract_feat=extract_feat, _skip_module_init=True, act_dtype=act_dtype)
        self.rw2 = RewightLayer(planes[0][1], planes[0][1], feat_depth['layer1'], 56)
        self.rw3 = RewightLayer(planes[1][1], planes[1][1], feat_depth['layer2'], 28)
        self.rw4 = RewightLayer(planes[2][1], planes[2][1], feat_depth['layer3'], 14)
        self.rw5 = RewightLayer(planes[3][1], planes[3][1], feat_depth['layer4'], 7)
        self.rw6 = RewightLayer(157, 157, feat_depth['conv5'], 7, pool=True)
        if self.isMixing:
            self.mix2 = MixingLayer(planes[0][1], learned=learnedMixing, index=0)
            self.mix3 = MixingLayer(planes[1][1], learned=learnedMixing, index=1)
            self.mix4 = MixingLayer(planes[2][1], learned=learnedMixing, index=2)
            self.mix5 = MixingLayer(planes[3][1], learned=learnedMixing, index=3)
        self.gauss = Gaussian(ratio=1)
        for m in self.modules():      # x3d_coarse.py:557-561: every Conv3d incl. SE and Grid Pool convs
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def replace_logits(self, n_classes):
        dev = self.fc1.weight.device
        self.fc2 = nn.Linear(2048, n_classes).to(dev)
        self.rw6 = RewightLayer(n_classes, n_classes, self.feat_depth['conv5'], 7, pool=True).to(dev)

    def forward(self, inp):
        x, feat, feat_masks, i, meta = inp
        tl = x.shape[2]
        x = self.layer1(self._stem(x))
        if self.act_dtype != torch.float32:          # 16-bit stem + layer 1: one widening pass, then the fp32 path (see __init__)
            x = (x.materialize() if isinstance(x, Deferred) else x).float()
        if self.t_pool == 'grid':
            x, gx = self.pool_1(x)
            GX = self.gauss([meta, feat_masks, gx, tl])
        else:
            if self.t_pool == 'stride':
                x = x[:, :, ::4].contiguous()
            elif self.t_pool in ('avg', 'max'):
                x = ops.time_pool(x.materialize() if isinstance(x, Deferred) else x, self.t_pool, 4)
            GX = self.gauss([meta, feat_masks, x, None])
        b2 = x.shape[0]
        levels = (('layer1', self.rw2), ('layer2', self.rw3), ('layer3', self.rw4), ('layer4', self.rw5))
        stages = (None, self.layer2, self.layer3, self.layer4)
        with _row_bank(b2, self._bias_groups(levels, feat, b2)):
            return self._fuse(x, feat, feat_masks, GX, gx if self.t_pool == 'grid' else None, b2, levels, stages)

    def _bias_groups(self, levels, feat, b2):
        """every bias vector the fusion branch turns into (b2, C) prologue rows during one forward (`_row_bank`)"""
        rws = [m for _, m in levels] + ([] if self.extract_feat else [self.rw6])
        groups = []
        for m in rws:
            groups += [[m.fc1.bias]] + ([[m.fc3.bias]] if m.g_channels is not None else [])
        for (k, m) in list(levels) + ([] if self.extract_feat else [('conv5', self.rw6)]):
            if feat[k].shape[0] == b2:               # asked for with the fine features' batch size (b2 / crops at multi-crop validation)
                groups.append([m.at1.bias])
        if self.isMixing:
            groups += [[m.fc2.bias for _, m in levels], [m.fc4.bias for _, m in levels]]
            for mix in (self.mix2, self.mix3, self.mix4, self.mix5):
                if mix.learned:
                    groups += [[mix.conv_at.bias], [mix.conv_at2.bias]]
        else:
            for _, m in levels:
                groups += [[m.fc2.bias], [m.fc4.bias]]
        return groups

    def _fuse(self, x, feat, feat_masks, GX, gx, b2, levels, stages):
        if self.isMixing:
            rw = [m.forward7(feat[k], b2, feat_masks, GX, True) for k, m in levels]
            rb, rs = MixingLayer.prepare([r[0] for r in rw]), MixingLayer.prepare([r[1] for r in rw])     # shared by the four mixing layers
            for li, mix in enumerate((self.mix2, self.mix3, self.mix4, self.mix5)):
                if stages[li] is not None:
                    x = stages[li](x)
                c7, m7 = mix.forward7(rb, rs, prepared=True)
                x = ops.film(x, m7, c7, x.shape[3] // FUSION_HW)
        else:
            for li, (k, m) in enumerate(levels):
                if stages[li] is not None:
                    x = stages[li](x)
                (x1, b1), (x2, b2_) = m.forward7(feat[k], b2, feat_masks, GX, False)
                one = _ones(b2, x1.shape[1], x.device)
                c7 = ops.affine_act(x1, one, _rows(b1, b2), ACT_NONE)
                m7 = ops.affine_act(x2, one, _rows(b2_, b2), ACT_SIGMOID)
                x = ops.film(x, m7, c7, x.shape[3] // FUSION_HW)
        x = self._head(x)
        if self.extract_feat:
            return x
        (x1, b1), (x2, b2_) = self.rw6.forward7(feat['conv5'], b2, feat_masks, GX, False)
        rw6 = x1.squeeze(4).squeeze(3) + b1.view(1, -1, 1)
        rw6_g = torch.sigmoid(x2.squeeze(4).squeeze(3) + b2_.view(1, -1, 1))
        x = x * rw6_g + rw6                                  # (B, n_classes, K): tiny
        if gx is not None:
            x = GridUnpool([x, gx, True])
            x = ops.time_resize(x, (x.shape[2] - 1) * 4)
        return x


def replace_logits(self, n_classes):
    self.fc2 = nn.Linear(2048, n_classes)


def generate_model(x3d_version, **kwargs):
    return ResNet(Bottleneck, get_blocks(x3d_version), get_inplanes(x3d_version), **kwargs)
