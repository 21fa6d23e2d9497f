"""GPU parity of the COARSE and JOINT streams at the shapes bench.py times (the fine stream's counterpart is
tests/test_hip_fullsize.py::test_bench_configuration_train_step_n8_t256):

  * the fusion gather kernels (register-tiled 4 x 13 (c, k) / 8 x 8 (t, k) blocks, csrc/fusion.hip) forward + every gradient against the fp64
    expression of x3d_coarse.py:209-223 at B = 8, C in {24, 432}, T' = 128, K in {17, 65}, 7 x 7 positions, crops in {1, 2}, masked fine steps;
  * BASELINE configs[3] per-GPU shard -- x3d_coarse on 8 x 3 x 64 x 224^2 + fine features T' = 128: eval logits of two of the eight clips against
    the CPU oracle (1e-3), Grid-Pool frame indices from the model's own CDF against the oracle's (interior knots exact), and -- with 8 BN splits,
    i.e. one clip per BN group -- the 8-clip train step against eight 1-clip steps (logits, summed parameter gradients);
  * BASELINE configs[4] per-GPU shard -- the joint two-stream step, fine tower on 8 x 128 frames feeding the coarse stream on the centre 64 --
    the same way.
"""
import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu
DEV = 'cuda'
DEPTH = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}


def _rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize('C,K,crops', [(24, 17, 1), (432, 17, 2), (432, 65, 1), (24, 65, 2)])
def test_fusion_gather_at_the_benchmarked_shapes(C, K, crops):
    """B = 8 videos x T' = 128 fine steps x 49 positions; K = 17 (coarse T = 64) / 65 (T = 256) pooled frames; rows with masked fine steps"""
    from cfn_hip import ops
    B, Tf, P = 8, 128, 49
    x, at_raw, bias = _rnd(1, B, C, Tf, P).abs(), _rnd(2, B, Tf, P), torch.tensor([0.3])
    GX, mask = _rnd(3, B * crops, Tf, K).abs(), torch.ones(B, Tf)
    mask[2, -37:] = 0          # a video shorter than the feature window
    mask[5, -1:] = 0
    c = [v.clone().double().requires_grad_(True) for v in (x, at_raw, bias, GX)]
    g = [v.clone().to(DEV).requires_grad_(True) for v in (x, at_raw, bias, GX)]

    def rep(v):
        return v.unsqueeze(1).repeat((1, crops) + (1,) * (v.dim() - 1)).view((B * crops,) + tuple(v.shape[1:]))
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    at = rep(torch.sigmoid(c[1] + c[2]))
    wgt = at.unsqueeze(2) * (c[3] * rep(mask.double()).unsqueeze(2)).unsqueeze(3)                       # B2 Tf K P
    zc = torch.einsum('bctp,btkp->bckp', rep(c[0]), wgt) / (wgt.sum(1) + 1e-6).unsqueeze(1)
    zg = ops.fusion_gather(g[0], g[1], g[2], g[3], mask.to(DEV), crops)
    assert tuple(zg.shape) == (B * crops, C, K, P)
    e_f = relerr(zg, zc)
    r = _rnd(5, *zc.shape)
    (zc * r.double()).sum().backward()
    (zg * r.to(DEV)).sum().backward()
    e_g = [relerr(a.grad, b.grad) for a, b in zip(g, c)]
    print('fusion_gather C=%d K=%d crops=%d: forward %.1e, gradients (x, attention, bias, GX) %s' % (C, K, crops, e_f, ['%.1e' % e for e in e_g]))
    assert e_f <= 1e-5
    assert all(e <= 1e-4 for e in e_g), e_g


def _coarse_inputs(seed, B, T, Tf):
    from oracle import spec
    x = spec.rand_input(seed, (B, 3, T, 224, 224))
    feat = {k: spec.rand_input(seed + 1 + i, (B, c, Tf, 7, 7), nonneg=True) for i, (k, c) in enumerate(DEPTH.items())}
    fm = torch.ones(B, Tf)
    meta = torch.zeros(B, 4, dtype=torch.int64)
    for b in range(B):
        valid = Tf - 5 * (b % 4)
        fm[b, valid:] = 0
        meta[b] = torch.tensor([(3 * b) % 7, T, valid, 1])
    return x, feat, fm, meta


def _coarse_model(splits):
    import x3d_coarse
    from oracle import spec
    m = x3d_coarse.generate_model('M', n_classes=400, feat_depth=DEPTH, task='loc', dropout=0.0, base_bn_splits=1, learnedMixing=True, isMixing=True,
                                  t_pool='grid')
    m.replace_logits(157)
    spec.fill_module_(m)
    m.to(DEV)
    if splits > 1:
        m.update_bn_splits_long_cycle(splits)      # every SubBatchNorm3d of the net (trunk AND Grid Pool): one clip per BN group
    m.rw6.dropout.p = 0.0
    return m


def _grad_errors(grads_ref, model):
    errs = {}
    for k, p in model.named_parameters():
        if p.grad is not None and k in grads_ref:
            errs[k] = float((grads_ref[k].double() - p.grad.double()).norm() / (p.grad.double().norm() + 1e-30))
    return errs


def _zero_gradient_by_construction(name):
    """biases whose output reaches nothing but train-mode batch norms as a per-channel constant: the Grid Pool convs in front of bn1 / bn2
    (x3d_coarse.py:362-366), the additive FiLM term of a stage-first block (conv1 -> bn1 and shortcut conv -> bn both remove it) and what feeds
    it linearly (mixN.conv_at.bias, the additive branch's rwN.fc2.bias).  Their true gradient is 0; what is computed is rounding noise, equally
    on both sides -- a relative error means nothing there.  Also left out of the per-tensor check: the ONE-element attention biases rwN.at2.bias --
    a uniform shift of the attention logits all but cancels in the normalised gather (x3d_coarse.py:219-223), the gradient is the small difference
    of large sums and moves by 6e-2 .. 3e-1 between two evaluations of the same 8-clip step (measured over four boxes)."""
    name = name.split('.', 1)[1] if name[:2] in ('f.', 'c.') else name
    return (name in ('pool_1.conv1.bias', 'pool_1.conv2.bias') or (name.startswith('mix') and name.endswith('.conv_at.bias'))
            or (name[:3] in ('rw2', 'rw3', 'rw4', 'rw5') and name.endswith('.fc2.bias')) or (name[:2] == 'rw' and name.endswith('.at2.bias')))


def _check_batch_equals_single_clips(tag, y8, grads8, y1s, model1):
    worst_y = max(float((yi - y8[i:i + 1]).abs().max()) for i, yi in enumerate(y1s))
    errs = _grad_errors(grads8, model1)
    norms = {k: float(p.grad.double().norm()) for k, p in model1.named_parameters() if p.grad is not None}
    assert set(errs) == set(norms)
    med_norm = sorted(norms.values())[len(norms) // 2]
    # left out of the per-tensor check: gradients that are zero by construction or cancel almost completely (see above), and gradients below 1e-3 of
    # the median gradient norm -- rounding noise on both sides.  They still count in the global figure below.
    noise = {k for k in errs if _zero_gradient_by_construction(k) or norms[k] < 1e-3 * med_norm}
    kept = {k: e for k, e in errs.items() if k not in noise}
    top = sorted(kept.items(), key=lambda kv: -kv[1])[:6]
    med = sorted(kept.values())[len(kept) // 2]
    glob = (sum((errs[k] * norms[k]) ** 2 for k in errs) / sum(norms[k] ** 2 for k in errs)) ** 0.5
    print('%s: 8-clip step (8 BN splits) vs eight 1-clip steps: logits max|diff| %.2e (max |logit| %.2f); gradients: all %d tensors together %.1e; '
          '%d tensors one by one: median %.1e, worst %s; %d noise-level gradients left out of that (norms <= %.1e against a median gradient norm of %.1e)'
          % (tag, worst_y, float(y8.abs().max()), len(errs), glob, len(kept), med, [(k, '%.1e' % e, '|g| %.1e' % norms[k]) for k, e in top], len(noise),
             max([norms[k] for k in noise] or [0.0]), med_norm))
    assert worst_y <= 2e-4 * max(float(y8.abs().max()), 1.0), worst_y
    # whole-net train-mode gradients: conditioned like the fine stream's -- two fp32 evaluations with different summation
    # orders agree to a few per cent per tensor (fine stream at T = 256: median <= 2e-2; here layers 2-4 see 17 frames per clip, and the joint
    # net chains two trunks: measured 2.4e-2 median on the coarse net, 5.4e-2 median / 5.5e-2 overall on the joint one); a wrong batch offset /
    # sample stride in any kernel shows as O(1) on the weights it touches and in the global figure
    # (per tensor: 0.22 is the worst seen over five boxes, on a squeeze-excite fc with 1e-2 of the median gradient norm)
    assert glob <= 0.1 and med <= 0.1 and all(e <= 0.5 for e in kept.values()), (glob, med, top)


def test_coarse_configuration_n8_t64_tf128():
    """BASELINE configs[3], one GPU's shard: 8 x 3 x 64 x 224^2 clips + fine features T' = 128 (what `bench.py --stream coarse` times)"""
    from cfn_hip import ops
    from oracle import spec, x3d_ref as R
    B, T, Tf = 8, 64, 128
    x, feat, fm, meta = _coarse_inputs(500, B, T, Tf)
    xd, featd, fmd, metad = x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), meta.to(DEV)
    # ---- eval mode against the CPU oracle: clips 0 and 7 of the batch of 8 ----
    m = _coarse_model(1).eval()
    with torch.no_grad():
        y = m([xd, featd, fmd, 0, metad])
        _, cdf = m.pool_1(m.layer1(m._stem(xd)))
    assert y.shape == (B, 157, T) and bool(torch.isfinite(y).all())
    K = cdf.shape[1]
    assert K == T // 4 + 1
    i_own, _ = ops.grid_time_index(cdf, T)
    sd = spec.procedural_fill(spec.coarse_keys('M', 157, 1))
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    for b in (0, B - 1):
        with torch.no_grad():
            yo, aux = R.x3d_coarse_forward(sd, [x[b:b + 1], {k: v[b:b + 1] for k, v in feat.items()}, fm[b:b + 1], 0, meta[b:b + 1]], 'M',
                                           training=False, return_aux=True)
        d = float((y[b:b + 1].cpu() - yo).abs().max())
        print('coarse 8x64 eval, clip %d: logits max|diff| vs oracle %.2e (max |logit| %.2f)' % (b, d, float(yo.abs().max())))
        assert d <= 1e-3
        assert float((cdf[b:b + 1].cpu() - aux['cdf']).abs().max()) <= 2e-6
        i_ref, _ = R.grid_sample_time_index(aux['cdf'], T)
        assert torch.equal(i_own[b:b + 1, :-1].cpu(), i_ref[:, :-1])                     # interior knots: exact
        assert int((i_own[b:b + 1, -1].cpu() - i_ref[:, -1]).abs().max()) <= 1           # last knot: T-2 (w=1) == T-1 (w=0), DESIGN section 2
    idx = i_own.cpu()
    assert bool((idx[:, 1:] >= idx[:, :-1]).all()) and int(idx.min()) >= 0 and int(idx.max()) <= T - 1
    del m, y
    # ---- train mode: the 8-clip step == eight 1-clip steps (one clip per BN group) ----
    m8 = _coarse_model(8).train(True)
    r = spec.rand_input(520, (B, 157, T)).to(DEV) / 100.0
    y8 = m8([xd, featd, fmd, 0, metad])
    (y8 * r).sum().backward()
    grads8 = {k: p.grad.detach().clone() for k, p in m8.named_parameters() if p.grad is not None}
    y8 = y8.detach()
    del m8
    m1 = _coarse_model(1).train(True)
    y1s = []
    for i in range(B):
        yi = m1([xd[i:i + 1].contiguous(), {k: v[i:i + 1].contiguous() for k, v in featd.items()}, fmd[i:i + 1], 0, metad[i:i + 1]])
        (yi * r[i:i + 1]).sum().backward()
        y1s.append(yi.detach())
    _check_batch_equals_single_clips('coarse 8x3x64x224^2 + T\'=128', y8, grads8, y1s, m1)


def test_joint_configuration_n8_fine128_coarse64():
    """BASELINE configs[4], one GPU's shard: fine tower on 8 x 3 x 128 x 224^2, coarse stream on the centre 64 frames (what
    `bench.py --stream joint` times)"""
    import train_joint as tj
    from oracle import spec, x3d_ref as R
    B, TF = 8, 128

    def nets(splits):
        fine, coarse = tj.build_models(DEV, dropout=0.0)
        spec.fill_module_(fine)
        spec.fill_module_(coarse)
        coarse.rw6.dropout.p = 0.0
        if splits > 1:
            fine.update_bn_splits_long_cycle(splits)
            coarse.update_bn_splits_long_cycle(splits)
        return fine, coarse
    clip = spec.rand_input(600, (B, 3, TF, 224, 224))
    cd = clip.to(DEV)
    fine, coarse = nets(1)
    fine.eval()
    coarse.eval()
    with torch.no_grad():
        logits, feat = tj.joint_forward(fine, coarse, cd)
    assert logits.shape == (B, 157, TF // 2) and bool(torch.isfinite(logits).all())
    sd_f = spec.procedural_fill(spec.fine_keys('M', 157, 1))
    sd_c = spec.procedural_fill(spec.coarse_keys('M', 157, 1))
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    for b in (1, B - 2):
        with torch.no_grad():
            feat_o = R.x3d_fine_forward(sd_f, clip[b:b + 1], 'M', training=False, global_tower=True)
            xc, s = tj.coarse_window(clip[b:b + 1])
            meta = torch.tensor([[s, TF // 2, TF, 1]], dtype=torch.int64)
            ref = R.x3d_coarse_forward(sd_c, [xc, feat_o, torch.ones(1, TF), 0, meta], 'M', training=False)
        for k in feat_o:
            assert float((feat[k][b:b + 1].cpu() - feat_o[k]).abs().max()) <= 1e-4 * max(float(feat_o[k].abs().max()), 1.0), k
        d = float((logits[b:b + 1].cpu() - ref).abs().max())
        print('joint 8x128/64 eval, clip %d: logits max|diff| vs oracle %.2e (max |logit| %.2f)' % (b, d, float(ref.abs().max())))
        assert d <= 1e-3
    del fine, coarse, logits, feat
    # ---- train mode: one 8-clip joint step == eight 1-clip joint steps, 8 BN splits in both nets ----
    fine8, coarse8 = nets(8)
    fine8.train(True)
    coarse8.train(True)
    r = spec.rand_input(620, (B, 157, TF // 2)).to(DEV) / 100.0
    y8, _ = tj.joint_forward(fine8, coarse8, cd)
    (y8 * r).sum().backward()
    g8 = {('f.' if m is fine8 else 'c.') + k: p.grad.detach().clone() for m in (fine8, coarse8) for k, p in m.named_parameters() if p.grad is not None}
    y8 = y8.detach()
    del fine8, coarse8
    fine1, coarse1 = nets(1)
    fine1.train(True)
    coarse1.train(True)
    y1s = []
    for i in range(B):
        yi, _ = tj.joint_forward(fine1, coarse1, cd[i:i + 1].contiguous())
        (yi * r[i:i + 1]).sum().backward()
        y1s.append(yi.detach())

    class _Both(object):                                   # named_parameters() over both nets with the prefixes used above
        @staticmethod
        def named_parameters():
            for pre, m in (('f.', fine1), ('c.', coarse1)):
                for k, p in m.named_parameters():
                    yield pre + k, p
    _check_batch_equals_single_clips('joint fine 8x3x128x224^2 -> coarse 64', y8, g8, y1s, _Both)
