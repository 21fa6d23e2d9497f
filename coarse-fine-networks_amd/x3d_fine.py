"""x3d_fine -- the Fine stream of Coarse-Fine Networks on MI355X (drop-in for the reference's
``x3d_fine.py``: same ``generate_model`` / ``ResNet.forward([x, masks])`` / ``replace_logits`` /
``aggregate_sub_bn_stats`` / ``update_bn_splits_long_cycle`` surface and the same state_dict keys).

Nothing here runs on stock ATen convolution / batch-norm kernels.  Every full-size tensor is
produced and consumed by the hand-written gfx950 kernels of ``cfn_hip`` (C ABI: include/cfn_hip.h):

  * convolutions return their *raw* output plus fp64 per-(sample, channel) ``sum`` / ``sum-of-squares``
    computed in the kernel epilogue;
  * SubBatchNorm3d, ReLU, Swish and the squeeze-excite gate are folded into a per-(sample, channel)
    affine ``A*x+B`` + activation that the *next* kernel applies while loading (``Deferred``), so
    BN/activation never cost a pass over HBM (the reference spends 3 eager passes per BN,
    x3d_fine.py:51-62);
  * the residual tail ``relu(bn3(.) + shortcut)`` is one kernel.

The (N,C)-sized statistics algebra (mean/var, running stats, SE FCs) is ordinary torch on tiny
tensors and is differentiated by autograd; reference semantics cited inline.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from cfn_hip import ops, ACT_NONE, ACT_RELU, ACT_SWISH

# CFN_USE_TORCH_OPS=1 (or x3d_fine.USE_TORCH_OPS = True): Bottleneck runs through the registered dispatcher operators
# torch.ops.cfn.* (cfn_hip/torchlib.py) with plain semantics -- the same kernels, visible to torch.compile / torch.export as opaque
# nodes; the cross-operator fusions of the default path (shortcut tokens, tail links, deferred prologues between blocks, batched
# gradient casts) are not expressible in functional operator signatures and are off in this mode.
USE_TORCH_OPS = os.environ.get('CFN_USE_TORCH_OPS', '0') == '1'


class Deferred(object):
    """A raw conv output whose normalisation/activation is still pending: the consumer kernel applies
    ``act(A[n,c]*raw + B[n,c])`` at load time."""
    __slots__ = ('raw', 'A', 'B', 'act')

    def __init__(self, raw, A, B, act):
        self.raw, self.A, self.B, self.act = raw, A, B, act

    @property
    def shape(self):
        return self.raw.shape

    def materialize(self):
        return ops.affine_act(self.raw, self.A, self.B, self.act)


def _unpack(x):
    if isinstance(x, Deferred):
        return x.raw, x.A, x.B, x.act
    return x, None, None, ACT_NONE


class SubBatchNorm3d(nn.Module):
    """Split batch-norm with one shared affine (reference x3d_fine.py:13-62, from SlowFast).

    ``fold`` turns the kernel-side statistics of a conv output into the (A, B) prologue of the next
    kernel; ``forward`` keeps the reference's tensor-in / tensor-out behaviour for stand-alone use."""

    def __init__(self, num_splits, **args):
        super(SubBatchNorm3d, self).__init__()
        self.num_splits = num_splits
        self.num_features = args['num_features']
        self.eps = args.get('eps', 1e-5)
        self.momentum = args.get('momentum', 0.1)
        if args.get('affine', True):
            self.affine = True
            self.weight = nn.Parameter(torch.ones(self.num_features))
            self.bias = nn.Parameter(torch.zeros(self.num_features))
        else:
            self.affine = False
        # buffer holders only (state_dict keys bn.* / split_bn.* as in the reference); never called
        self.bn = nn.BatchNorm3d(self.num_features, eps=self.eps, momentum=self.momentum, affine=False)
        self.split_bn = nn.BatchNorm3d(self.num_features * self.num_splits, eps=self.eps, momentum=self.momentum,
                                       affine=False)

    def aggregate_stats(self):
        """x3d_fine.py:30-49: mean of split means; mean of split vars + variance of split means."""
        if self.split_bn.track_running_stats:
            n = self.num_splits
            means = self.split_bn.running_mean.view(n, -1)
            mean = means.sum(0) / n
            var = self.split_bn.running_var.view(n, -1).sum(0) / n + ((means - mean) ** 2).sum(0) / n
            self.bn.running_mean.data = mean.detach()
            self.bn.running_var.data = var.detach()

    def fold(self, s, q, count, n, se=None):
        """(A, B) fp32 (n, C) such that SubBN(y)[n,c,...] = A[n,c]*y + B[n,c]  (one fused HIP kernel).

        training: s, q = fp64 (n, C) per-sample sums of y and y*y over ``count`` positions each; statistics
        are taken per split group exactly like ``x.view(n//S, c*S, ...)`` (x3d_fine.py:52-57) and the
        split_bn running statistics are updated like nn.BatchNorm3d does.  eval: running statistics of
        ``self.bn``.  ``se`` = (fc1.weight, fc1.bias, fc2.weight, fc2.bias) additionally folds the
        squeeze-excite gate of x3d_fine.py:157-163 into (A, B) (needs ``s`` in eval mode too)."""
        if self.training:
            bufs = (self.split_bn.running_mean, self.split_bn.running_var, self.split_bn.num_batches_tracked)
        else:
            bufs = (self.bn.running_mean, self.bn.running_var, self.bn.num_batches_tracked)
        gamma, beta = (self.weight, self.bias) if self.affine else (None, None)
        return ops.bn_fold(s, q, gamma, beta, bufs, self.training, n, self.num_features, self.num_splits, count, self.eps,
                           self.momentum, se=se, pool_count=count)

    def fold_op(self, s, q, count, n, se=None):
        """`fold` through the functional dispatcher operator torch.ops.cfn.bn_fold: the operator returns the updated running
        statistics, which are copied into the buffers here"""
        tr = self.training
        bn = self.split_bn if tr else self.bn
        gamma, beta = (self.weight, self.bias) if self.affine else (None, None)
        w1, b1, w2, b2 = se if se is not None else (None, None, None, None)
        out = torch.ops.cfn.bn_fold(s, q, gamma, beta, bn.running_mean, bn.running_var, bn.num_batches_tracked, tr, n,
                                    self.num_features, self.num_splits, float(count), self.eps, self.momentum, w1, b1, w2, b2, float(count))
        if tr:
            with torch.no_grad():
                bn.running_mean.copy_(out[9])
                bn.running_var.copy_(out[10])
                bn.num_batches_tracked.copy_(out[11])
        return out[0], out[1]

    def forward(self, x):
        n = x.shape[0]
        s = q = None
        if self.training:
            s, q = ops.channel_stats(x)
        A, B = self.fold(s, q, x[0, 0].numel(), n)
        return ops.affine_act(x, A, B, ACT_NONE)


class Swish(nn.Module):
    """x*sigmoid(x) (x3d_fine.py:65-86).  Inside Bottleneck it is fused into conv3's prologue; this
    module form is kept for API parity."""

    def forward(self, x):
        n, c = x.shape[:2]
        one = torch.ones(n, c, device=x.device)
        return ops.affine_act(x, one, torch.zeros_like(one), ACT_SWISH)


def conv3x3x3(in_planes, out_planes, stride=1, t_downsample=False):
    """depthwise 3x3x3 (x3d_fine.py:89-97); t_downsample: the stride also applies along t"""
    return nn.Conv3d(in_planes, out_planes, kernel_size=3, stride=stride if t_downsample else (1, stride, stride), padding=1,
                     bias=False, groups=in_planes)


def conv1x1x1(in_planes, out_planes, stride=1, t_downsample=False):
    return nn.Conv3d(in_planes, out_planes, kernel_size=1, stride=stride if t_downsample else (1, stride, stride), bias=False)


def _every(x, s):
    """x[:, :, ::s] as a dense tensor (temporal stride of the never-default t_downsample option: the kernels stride H and W
    only; the frame selection is an indexing op around them)"""
    return x if s == 1 else x[:, :, ::s].contiguous()


def _count(y):
    return y.shape[2] * y.shape[3] * y.shape[4]


class Bottleneck(nn.Module):
    """pw -> BN/ReLU -> dw3x3x3 -> BN [-> SE] -> Swish -> pw -> BN -> (+shortcut) -> ReLU
    (x3d_fine.py:108-175).  The nn.Conv3d members only hold parameters (names, shapes, init)."""

    def __init__(self, in_planes, planes, stride=1, downsample=None, index=0, base_bn_splits=8, t_downsample=False):
        super(Bottleneck, self).__init__()
        self.index = index
        self.base_bn_splits = base_bn_splits
        self.conv1 = conv1x1x1(in_planes, planes[0])
        self.bn1 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=planes[0], affine=True)
        self.conv2 = conv3x3x3(planes[0], planes[0], stride, t_downsample=t_downsample)
        self.bn2 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=planes[0], affine=True)
        self.conv3 = conv1x1x1(planes[0], planes[1], t_downsample=t_downsample)
        self.bn3 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=planes[1], affine=True)
        self.swish = Swish()
        self.relu = nn.ReLU(inplace=True)
        if self.index % 2 == 0:
            width = self.round_width(planes[0])
            self.global_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
            self.fc1 = nn.Conv3d(planes[0], width, kernel_size=1, stride=1)
            self.fc2 = nn.Conv3d(width, planes[0], kernel_size=1, stride=1)
            self.sigmoid = nn.Sigmoid()
        self.downsample = downsample
        self.stride = stride
        self.t_stride = stride if t_downsample else 1
        # set by ResNet._make_layer when the next module is an identity-shortcut Bottleneck: the output is then
        # returned as a pair so that its two gradients (conv1 data gradient, residual) meet inside the tail kernel
        self.split_out = False

    def round_width(self, width, multiplier=0.0625, min_width=8, divisor=8):
        if not multiplier:
            return width
        width *= multiplier
        min_width = min_width or divisor
        width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
        if width_out < 0.9 * width:
            width_out += divisor
        return int(width_out)

    def _forward_torch_ops(self, x):
        """the block on torch.ops.cfn.* (plain tensors in and out; x3d_fine.py:146-175)"""
        import cfn_hip.torchlib  # noqa: F401  (registers the operators)
        T = torch.ops.cfn
        if isinstance(x, tuple):
            x = x[0]
        if isinstance(x, Deferred):
            x = x.materialize()
        if self.t_stride != 1 or (self.downsample is not None and not isinstance(self.downsample, nn.Sequential)):
            raise NotImplementedError('CFN_USE_TORCH_OPS: t_downsample / shortcut type A run on the default path only')
        n = x.shape[0]
        has_se = self.index % 2 == 0
        y1, s1, q1 = T.pwconv(x, self.conv1.weight, None, None, ACT_NONE, 1)
        A1, B1 = self.bn1.fold_op(s1, q1, _count(y1), n)
        y2, s2, q2 = T.dwconv3d(y1, self.conv2.weight, A1, B1, ACT_RELU, self.stride)
        se = (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias) if has_se else None
        A2, B2 = self.bn2.fold_op(s2, q2, _count(y2), n, se=se)
        y3, s3, q3 = T.pwconv(y2, self.conv3.weight, A2, B2, ACT_SWISH, 1)
        A3, B3 = self.bn3.fold_op(s3, q3, _count(y3), n)
        if self.downsample is not None:
            yd, sd, qd = T.pwconv(x, self.downsample[0].weight, None, None, ACT_NONE, self.stride)
            Ad, Bd = self.downsample[1].fold_op(sd, qd, _count(yd), n)
            return T.bn_add_relu(y3, A3, B3, yd, Ad, Bd)
        return T.bn_add_relu(y3, A3, B3, x)

    def forward(self, x):
        if USE_TORCH_OPS:
            return self._forward_torch_ops(x)
        # a (x_for_conv1, x_for_residual) pair: the previous block handed out its output twice (see split_out below)
        x, x_res = x if isinstance(x, tuple) else (x, x)
        xr, xa, xb, xact = _unpack(x)
        n = xr.shape[0]
        tr = self.training
        has_se = self.index % 2 == 0

        ts = self.t_stride
        # stage-first block: conv1 and the strided shortcut conv read the same input; their data gradients are fused
        # (same frames only: with a temporal stride the shortcut sees every ts-th frame and the plain path is used)
        tok = ops.ShortcutToken() if (isinstance(self.downsample, nn.Sequential) and self.stride > 1 and ts == 1) else None
        y1, s1, q1 = ops.pwconv(xr, self.conv1.weight, xa, xb, xact, 1, stats=tr, token=tok, role='main')
        A1, B1 = self.bn1.fold(s1, q1, _count(y1), n)
        if ts == 1:
            y2, s2, q2 = ops.dwconv3d(y1, self.conv2.weight, A1, B1, ACT_RELU, self.stride, stats=tr or has_se)
        else:   # t_downsample (x3d_fine.py:93): a kernel-3 / pad-1 conv at temporal stride ts = every ts-th frame of the stride-1 result
            y2f, _, _ = ops.dwconv3d(y1, self.conv2.weight, A1, B1, ACT_RELU, self.stride, stats=False)
            y2 = _every(y2f, ts)
            s2, q2 = ops.channel_stats(y2) if (tr or has_se) else (None, None)
        # bn2 (+ SE: the global average of bn2(y2) is the bn2 affine of the per-sample mean of y2, x3d_fine.py:157-163)
        se = (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias) if has_se else None
        A2, B2 = self.bn2.fold(s2, q2, _count(y2), n, se=se)
        # the tail's backward writes one unscaled gradient tensor; conv3 (and the shortcut conv) apply bn3's (the
        # shortcut bn's) per-(n,c) factor when they load it
        link = ops.TailLink() if torch.is_grad_enabled() else None
        y3, s3, q3 = ops.pwconv(y2, self.conv3.weight, A2, B2, ACT_SWISH, 1, stats=tr, tail=link, tail_role='y')
        A3, B3 = self.bn3.fold(s3, q3, _count(y3), n)

        if self.downsample is not None:
            if not isinstance(self.downsample, nn.Sequential):
                # shortcut type 'A' (x3d_fine.py:266-275): avg_pool3d(kernel 1, stride s) = every s-th sample along t, h, w, zero
                # channels appended, and -- `out.data` -- cut out of the autograd graph
                _, planes_out, s_ = self.downsample
                xm = x_res.materialize() if isinstance(x_res, Deferred) else x_res
                res = xm.detach()[:, :, ::s_, ::s_, ::s_]
                if planes_out > res.shape[1]:
                    res = torch.cat([res, res.new_zeros((res.shape[0], planes_out - res.shape[1]) + tuple(res.shape[2:]))], dim=1)
                return ops.bn_add_relu(y3, A3, B3, res.contiguous(), split=self.split_out, link=link)
            yd, sd, qd = ops.pwconv(_every(xr, ts), self.downsample[0].weight, xa, xb, xact, self.stride, stats=tr, token=tok, role='short',
                                    tail=link, tail_role='res')
            Ad, Bd = self.downsample[1].fold(sd, qd, _count(yd), n)
            return ops.bn_add_relu(y3, A3, B3, yd, Ad, Bd, split=self.split_out, link=link)
        res = x_res.materialize() if isinstance(x_res, Deferred) else x_res
        return ops.bn_add_relu(y3, A3, B3, res, split=self.split_out, link=link)


class ResNet(nn.Module):
    """X3D trunk + localisation / classification head (x3d_fine.py:179-382)."""

    def __init__(self, block, layers, block_inplanes, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1,
                 shortcut_type='B', widen_factor=1.0, dropout=0.5, n_classes=400, base_bn_splits=8, task='class',
                 extract_feat=False, global_tower=False, t_downsample=False, aux_losses=None, _skip_module_init=False,
                 act_dtype=None):
        if not _skip_module_init:      # x3d_coarse registers pool_1 before the trunk, like the reference
            super(ResNet, self).__init__()
        block_inplanes = [(int(x * widen_factor), int(y * widen_factor)) for x, y in block_inplanes]
        self.index = 0
        self.base_bn_splits = base_bn_splits
        self.task = task
        self.extract_feat = extract_feat
        self.global_tower = global_tower
        self.t_downsample = t_downsample
        # storage type of the activations and their gradients in HBM (an addition to the reference's signature): None /
        # 'f32' = fp32 (the reference's precision), 'bf16' = bf16 storage + bf16 MFMA with fp32 accumulation, fp32 master
        # weights, fp32/fp64 statistics (BASELINE configs[1]).  The clip, the stem conv output and everything after the
        # spatial pooling of the head stay fp32.
        self.act_dtype = {None: torch.float32, 'f32': torch.float32, 'fp32': torch.float32, torch.float32: torch.float32,
                          'bf16': torch.bfloat16, torch.bfloat16: torch.bfloat16,
                          # 'fp16': IEEE-half storage + v_mfma_f32_32x32x16_f16 (BASELINE configs[4]); activation gradients then need a loss scale
                          'fp16': torch.float16, 'f16': torch.float16, torch.float16: torch.float16}[act_dtype]
        self.in_planes = block_inplanes[0][1]

        self.conv1_s = nn.Conv3d(n_input_channels, self.in_planes, kernel_size=(1, 3, 3), stride=(1, 2, 2),
                                 padding=(0, 1, 1), bias=False)
        self.conv1_t = nn.Conv3d(self.in_planes, self.in_planes, kernel_size=(5, 1, 1), stride=(1, 1, 1),
                                 padding=(2, 0, 0), bias=False, groups=self.in_planes)
        self.bn1 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=self.in_planes, affine=True)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, block_inplanes[0], layers[0], shortcut_type, stride=2)
        self.layer2 = self._make_layer(block, block_inplanes[1], layers[1], shortcut_type, stride=2)
        self.layer3 = self._make_layer(block, block_inplanes[2], layers[2], shortcut_type, stride=2)
        self.layer4 = self._make_layer(block, block_inplanes[3], layers[3], shortcut_type, stride=2)
        self.conv5 = nn.Conv3d(block_inplanes[3][1], block_inplanes[3][0], kernel_size=(1, 1, 1), stride=(1, 1, 1),
                               padding=(0, 0, 0), bias=False)
        self.bn5 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=block_inplanes[3][0], affine=True)
        if task == 'class':
            self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        elif task == 'loc':
            self.avgpool = nn.AdaptiveAvgPool3d((None, 1, 1))
        self.fc1 = nn.Conv3d(block_inplanes[3][0], 2048, bias=False, kernel_size=1, stride=1)
        self.fc2 = nn.Linear(2048, n_classes)
        self.dropout = nn.Dropout(dropout)

        for m in self.modules():   # x3d_fine.py:260-264
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make_layer(self, block, planes, blocks, shortcut_type, stride=1):
        downsample = None
        if stride != 1 or self.in_planes != planes[1]:
            if shortcut_type == 'A':
                # parameter-free shortcut; the reference's own 'A' strides t as well, so it only fits t_downsample=True
                # (x3d_fine.py:266-275: with the default temporal stride 1 its `out += residual` fails on the T axis)
                downsample = ('A', planes[1], stride)
            else:
                downsample = nn.Sequential(
                    conv1x1x1(self.in_planes, planes[1], stride, t_downsample=self.t_downsample),
                    SubBatchNorm3d(num_splits=self.base_bn_splits, num_features=planes[1], affine=True))
        layers = [block(in_planes=self.in_planes, planes=planes, stride=stride, downsample=downsample,
                        index=self.index, base_bn_splits=self.base_bn_splits, t_downsample=self.t_downsample)]
        self.in_planes = planes[1]
        self.index += 1
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, planes, index=self.index, base_bn_splits=self.base_bn_splits,
                                t_downsample=self.t_downsample))
            self.index += 1
        self.index = 0
        for blk in layers[:-1]:
            blk.split_out = True
        return nn.Sequential(*layers)

    def replace_logits(self, n_classes):
        self.fc2 = nn.Linear(2048, n_classes).to(self.fc1.weight.device)

    def update_bn_splits_long_cycle(self, long_cycle_bn_scale):
        for m in self.modules():
            if isinstance(m, SubBatchNorm3d):
                m.num_splits = self.base_bn_splits * long_cycle_bn_scale
                m.split_bn = nn.BatchNorm3d(num_features=m.num_features * m.num_splits, affine=False).to(m.weight.device)
        return self.base_bn_splits * long_cycle_bn_scale

    def aggregate_sub_bn_stats(self):
        count = 0
        for m in self.modules():
            if isinstance(m, SubBatchNorm3d):
                m.aggregate_stats()
                count += 1
        return count

    # -- pieces shared with x3d_coarse ---------------------------------------------------------------
    def _stem(self, x):
        """conv1_s -> conv1_t -> bn1 -> relu (x3d_fine.py:334-337); bn1+relu stay deferred."""
        y = ops.stem_conv(x, self.conv1_s.weight)
        y, s, q = ops.dwconv_t5(y, self.conv1_t.weight, stats=self.training, out_dtype=self.act_dtype)
        A, B = self.bn1.fold(s, q, _count(y), y.shape[0])
        return Deferred(y, A, B, ACT_RELU)

    def _head(self, x):
        """conv5 -> bn5 -> relu -> avgpool(H,W) -> fc1 -> relu -> dropout -> fc2 (x3d_fine.py:356-380)."""
        n = x.shape[0]
        y5, s5, q5 = ops.pwconv(x, self.conv5.weight, stats=self.training)
        A5, B5 = self.bn5.fold(s5, q5, _count(y5), n)
        if self.task == 'class':
            pooled = ops.pool_hw(y5, 1, 1, A5, B5, ACT_RELU).mean(dim=2, keepdim=True)
        else:
            pooled = ops.pool_hw(y5, 1, 1, A5, B5, ACT_RELU)           # (N, C, T, 1, 1)
        if self.extract_feat:
            return pooled
        f1, _, _ = ops.pwconv(pooled, self.fc1.weight, stats=False)
        one = torch.ones(n, f1.shape[1], device=f1.device)
        z = ops.affine_act(f1, one, torch.zeros_like(one), ACT_RELU)
        z = self.dropout(z)
        out, _, _ = ops.pwconv(z, self.fc2.weight.view(self.fc2.weight.shape[0], -1, 1, 1, 1), stats=False)
        out = out.squeeze(4).squeeze(3) + self.fc2.bias.view(1, -1, 1)  # (N, n_classes, T or 1)
        return out

    def forward(self, inp):
        x, masks = inp
        x = self._stem(x)
        feat_g = {}
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4), start=1):
            x = layer(x)
            if self.global_tower:
                feat_g['layer%d' % i] = ops.pool_hw(x, 7, 7)
        if self.global_tower:
            n = x.shape[0]
            y5, s5, q5 = ops.pwconv(x, self.conv5.weight, stats=self.training)
            A5, B5 = self.bn5.fold(s5, q5, _count(y5), n)
            feat_g['conv5'] = ops.pool_hw(y5, 7, 7, A5, B5, ACT_RELU)
            return feat_g, masks
        return self._head(x)


def replace_logits(self, n_classes):
    self.fc2 = nn.Linear(2048, n_classes)


def get_inplanes(version):
    planes = {'S': [(54, 24), (108, 48), (216, 96), (432, 192)],
              'M': [(54, 24), (108, 48), (216, 96), (432, 192)],
              'XL': [(72, 32), (162, 72), (306, 136), (630, 280)]}
    return planes[version]


def get_blocks(version):
    blocks = {'S': [3, 5, 11, 7], 'M': [3, 5, 11, 7], 'XL': [5, 10, 25, 15]}
    return blocks[version]


def generate_model(x3d_version, **kwargs):
    return ResNet(Bottleneck, get_blocks(x3d_version), get_inplanes(x3d_version), **kwargs)
