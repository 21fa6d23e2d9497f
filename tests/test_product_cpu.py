"""Host-side product code pinned to outputs of the reference (CPU, no HIP calls): the loss of both training scripts, the
two batch builders, the per-GPU hand-off structures.  Fixtures: tests/golden/make_golden.py (loss_ap, loss_multicrop,
collate_fine, collate_coarse)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, t, maxdiff


@pytest.mark.parametrize('ac', [1, 0])
def test_product_detection_loss_matches_reference(ac):
    """train_fine.detection_loss (align_corners=True, train_fine.py:199-213) and train_coarse_fineFEAT.detection_loss
    (no align_corners, train_coarse_fineFEAT.py:226-240) against the losses the reference's own expressions gave"""
    import train_fine
    import train_coarse_fineFEAT as tc
    z = load_golden('loss_ap')
    lg, labels, masks = t(z['logits']), t(z['labels']), t(z['masks'])
    if ac:
        cls, loc, _ = train_fine.detection_loss(lg, labels, masks, True)
    else:
        cls, loc, _ = tc.detection_loss(lg, labels, masks)
    assert abs(float(cls) - float(z['cls_%d' % ac])) <= 1e-6
    assert abs(float(loc) - float(z['loc_%d' % ac])) <= 1e-6


@pytest.mark.parametrize('ac', [1, 0])
def test_product_detection_loss_multicrop(ac):
    """validation branch: n crops per video, max over crops (train_fine.py:204-207)"""
    import train_fine
    import train_coarse_fineFEAT as tc
    z = load_golden('loss_multicrop')
    lg, labels, masks, n = t(z['logits']), t(z['labels']), t(z['masks']), int(z['crops'])
    if ac:
        cls, loc, probs = train_fine.detection_loss(lg, labels, masks, True, crops=n, local_norm=True)
    else:
        cls, loc, probs = tc.detection_loss(lg, labels, masks, crops=n, local_norm=True)
    assert probs.shape == labels.shape
    assert abs(float(cls) - float(z['cls_%d' % ac])) <= 1e-6 and abs(float(loc) - float(z['loc_%d' % ac])) <= 1e-6
    assert maxdiff(probs[:, ::13], z['probs_%d' % ac]) <= 1e-6


def test_fine_collate_equals_reference_mt_collate_fn():
    import collate
    z = load_golden('collate_fine')
    batch = [(z['in%d_clips' % i], z['in%d_label' % i], 'vid%d' % i) for i in range(3)]
    clips, label, mask, vids = collate.fine_collate(batch)
    for mine, key in ((clips, 'clips'), (label, 'label'), (mask, 'mask')):
        assert mine.dtype == torch.float32 and tuple(mine.shape) == z[key].shape and torch.equal(mine, t(z[key])), key
    assert list(vids) == json.loads(str(z['vids']))


def test_coarse_collate_equals_reference_mt_collate_fn():
    import collate
    z = load_golden('collate_coarse')
    keys = ('layer1', 'conv5')
    batch = [(z['in%d_clips' % i], z['in%d_label' % i], {k: z['in%d_feat_%s' % (i, k)] for k in keys}, z['in%d_meta' % i],
              'vid%d' % i, 10.5 + i) for i in range(3)]
    clips, label, mask, feat, fmask, meta, vids, dur = collate.coarse_collate(batch)
    for mine, key in ((clips, 'clips'), (label, 'label'), (mask, 'mask'), (fmask, 'fmask'), (meta, 'meta'), (dur, 'dur')):
        ref = t(z[key])
        assert mine.dtype == ref.dtype and torch.equal(mine, ref), key
    for k in keys:
        assert torch.equal(feat[k], t(z['feat_' + k])), k
    assert fmask.shape[1] == 128 and float(fmask[0].sum()) == 100.0       # capped at 128, sample 0 is 100 frames long
    assert list(vids) == json.loads(str(z['vids']))


# ---------------------------------------------------------------------------------------------------------------------------
# checkpoint compatibility and learning-rate schedule of the two entry points
# ---------------------------------------------------------------------------------------------------------------------------
def _reference_checkpoint(tmp_path, which):
    """a checkpoint file laid out like models/x3d_multigrid_kinetics_fb_pretrained.pt: the reference's own key set and shapes
    for the 400-class Kinetics X3D-M (tests/golden/state_keys.npz, captured from the reference's state_dict), values filled
    procedurally by key name"""
    import os
    from conftest import GOLDEN
    from oracle import spec
    z = np.load(os.path.join(GOLDEN, 'state_keys.npz'))
    shapes = {k: tuple(s) for k, s in json.loads(str(z[which]))}
    # the Kinetics checkpoint holds the plain X3D trunk with a 400-way head: fc2 is (400, 2048); the coarse-only modules
    # (rw*, mix*, pool_1) are not in it
    shapes = {k: ((400,) + s[1:] if k.startswith('fc2.') else s) for k, s in shapes.items()
              if not any(p in k for p in ('rw', 'mix', 'pool_1'))}
    sd = spec.procedural_fill(shapes)
    path = str(tmp_path / 'kinetics_like.pt')
    torch.save({'model_state_dict': sd}, path)
    return path, sd


@pytest.mark.parametrize('which', ['fine', 'coarse'])
def test_reference_format_checkpoint_loads_through_build_model(tmp_path, which):
    """train_fine.py:104-110 / train_coarse_fineFEAT.py:112-118: state.update(ckpt['model_state_dict']) on a 400-class model,
    then replace_logits(157): every trunk tensor comes from the checkpoint, fc2 is re-drawn at (157, 2048), the modules the
    checkpoint does not know keep their initial values"""
    import train_fine
    import train_coarse_fineFEAT as tc
    path, sd = _reference_checkpoint(tmp_path, which)
    build = train_fine.build_model if which == 'fine' else tc.build_model
    net = build(torch.device('cpu'), pretrained=path)
    mine = net.state_dict()
    assert tuple(mine['fc2.weight'].shape) == (157, 2048) and tuple(mine['fc2.bias'].shape) == (157,)
    checked = 0
    for k, v in sd.items():
        if k.startswith('fc2.'):
            continue
        assert k in mine and torch.equal(mine[k], v), k
        checked += 1
    assert checked >= 800
    if which == 'coarse':
        assert any('rw' in k for k in mine) and any('mix' in k for k in mine)
    # the run() checkpoints (every 1000 steps) hold net.state_dict(): a fresh model loads them strictly (resume path,
    # train_fine.py:112-114)
    again = build(torch.device('cpu'), pretrained=None)
    again.load_state_dict({k: v.clone() for k, v in mine.items()})
    for k, v in again.state_dict().items():
        assert torch.equal(v, mine[k]), k
    with pytest.raises(FileNotFoundError):
        build(torch.device('cpu'), pretrained=str(tmp_path / 'missing.pt'))


@pytest.mark.parametrize('which', ['fine', 'coarse'])
def test_learning_rate_steps_once_per_val_phase(which):
    """train_fine.py:131,256 -- MultiStepLR([15, 20, 25]) stepped after every VAL phase (4 train epochs per val phase);
    train_coarse_fineFEAT.py:147,296 -- MultiStepLR([15, 25, 35]), 2 train epochs per val phase, the rw / mix group at 10x.
    Empty loaders: the phase / epoch / scheduler bookkeeping of run() without a single forward."""
    import train_fine
    import train_coarse_fineFEAT as tc
    mod, milestones, per_val = (train_fine, [15, 20, 25], 4) if which == 'fine' else (tc, [15, 25, 35], 2)
    seen = []

    def hook(phase, epochs, opt):
        seen.append((phase, epochs, [g['lr'] for g in opt.param_groups]))

    kw = dict(csv_path=None) if which == 'coarse' else {}
    mod.run(init_lr=0.02, max_epochs=per_val * 40, dataloaders={'train': [], 'val': []}, pretrained=None, log=lambda *_: None,
            phase_hook=hook, **kw)
    vals = [s for s in seen if s[0] == 'val']
    assert len(vals) == 40 and [s[1] for s in vals] == [per_val * (i + 1) for i in range(40)]     # epochs count TRAIN phases
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.02)
    ref_sched = torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones)
    for k, (_, _, lrs) in enumerate(vals):
        ref_sched.step()                                     # k + 1 val phases so far
        want = ref_opt.param_groups[0]['lr']
        assert abs(lrs[0] - want) <= 1e-12, (k, lrs, want)
        if which == 'coarse':
            assert len(lrs) == 2 and abs(lrs[1] - 10 * want) <= 1e-12      # the 10x group keeps its ratio through the decays
    trains = [s for s in seen if s[0] == 'train']
    assert len(trains) == per_val * 40 and abs(trains[0][2][0] - 0.02) <= 1e-12
    assert abs(vals[14][2][0] - 0.002) <= 1e-12 and abs(vals[13][2][0] - 0.02) <= 1e-12     # first decay after the 15th val phase


def test_loss_scaler_unscales_and_survives_an_overflow():
    """train_fine.LossScaler (fp16 activation path): clean gradients are divided by the scale; ONE inf / nan anywhere zeroes
    every gradient (the step that follows moves nothing but momentum / weight decay) and halves the scale -- all on tensors, no .item()"""
    import train_fine
    sc = train_fine.LossScaler('cpu', init=1024.0, interval=2)
    ps = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2)), torch.nn.Parameter(torch.ones(1))]
    for p in ps[:2]:
        p.grad = torch.full_like(p, 2048.0)
    assert float(sc.scale_loss(torch.tensor(0.5))) == 512.0
    sc.unscale_(ps)                                   # (a parameter without a gradient is skipped)
    assert all(bool((p.grad == 2.0).all()) for p in ps[:2]) and float(sc.scale) == 1024.0 and float(sc.found_inf) == 0.0
    for p in ps[:2]:
        p.grad = torch.full_like(p, 2048.0)
    ps[1].grad[0, 0] = float('inf')
    ps[0].grad[1] = float('nan')
    sc.unscale_(ps)
    assert all(bool((p.grad == 0.0).all()) for p in ps[:2]) and float(sc.scale) == 512.0 and float(sc.found_inf) == 1.0
    for _ in range(2):                                # `interval` clean steps: the scale grows back
        for p in ps[:2]:
            p.grad = torch.full_like(p, 512.0)
        sc.unscale_(ps)
    assert float(sc.scale) == 1024.0 and bool((ps[0].grad == 1.0).all())
    # the plain-factor and no-op spellings of unscale_grads
    ps[0].grad = torch.full((3,), 8.0)
    train_fine.unscale_grads(ps, 4.0)
    assert bool((ps[0].grad == 2.0).all())
    train_fine.unscale_grads(ps, None)
    train_fine.unscale_grads(ps, 1.0)
    assert bool((ps[0].grad == 2.0).all())
    net = torch.nn.Linear(2, 2)
    assert train_fine.loss_scaler(net) is None and train_fine.loss_scale(net) == 1.0


def test_row_bank_equals_single_row_ops_forward_and_backward():
    """x3d_coarse._row_bank (round 6): every bias of the fusion branch becomes its (n, C) fp64 prologue rows in ONE autograd Function.  The
    blocks and the gradients must equal what one `_Rows64` per vector gives; a request with another n or an unknown vector falls back to
    `_Rows64`; a block nobody used contributes an exactly-zero gradient; banks nest and are per thread."""
    import threading
    import x3d_coarse as xc
    g = torch.Generator().manual_seed(3)
    sizes = (24, 48, 157, 5, 7)
    vs = [torch.randn(c, generator=g, requires_grad=True) for c in sizes]
    ref = [v.detach().clone().requires_grad_(True) for v in vs]
    n = 3
    groups = [[vs[0]], [vs[1], vs[2]], [vs[3]], [vs[4]], [vs[0]]]          # (a repeated group is stored once)
    with xc._row_bank(n, groups) as bank:
        assert len(bank.groups) == 4
        a, b, c = xc._rows(vs[0], n), xc._rows_of([vs[1], vs[2]], n), xc._rows(vs[3], n)
        other_n = xc._rows(vs[4], 2)                                      # n differs: single op
        assert xc._rows(vs[0], n) is a                                    # one block per group, handed out again
        with xc._row_bank(n, [[vs[3]]]):                                   # nested: the inner bank answers, the outer one is restored
            inner = xc._rows(vs[3], n)
        assert inner is not c and xc._rows(vs[3], n) is c
        seen = []
        th = threading.Thread(target=lambda: seen.append(getattr(xc._BANK, 'cur', None)))
        th.start(); th.join()
        assert seen == [None]
    assert getattr(xc._BANK, 'cur', None) is None
    ra, rb, rc = xc._Rows64.apply(ref[0], n), xc._Rows64.apply(torch.cat([ref[1], ref[2]]), n), xc._Rows64.apply(ref[3], n)
    for got, want in ((a, ra), (b, rb), (c, rc)):
        assert got.dtype == torch.float64 and got.is_contiguous() and got.data_ptr() % 8 == 0 and torch.equal(got, want)
    assert (b.data_ptr() - a.data_ptr()) % 256 == 0 and (c.data_ptr() - a.data_ptr()) % 256 == 0    # each block aligned like a tensor of its own
    assert other_n.shape == (2, 7)
    wa, wb = torch.randn(a.shape, generator=g, dtype=torch.float64), torch.randn(b.shape, generator=g, dtype=torch.float64)
    ((a * wa).sum() + (b * wb).sum() + (other_n * 2.0).sum()).backward()
    ((ra * wa).sum() + (rb * wb).sum()).backward()
    for got, want in zip(vs[:3], ref[:3]):       # (the bank sums in fp64 and narrows once; `_Rows64` narrows first)
        assert maxdiff(got.grad, want.grad) <= 2e-6
    assert vs[0].grad.dtype == torch.float32 and vs[2].grad.shape == (157,)
    assert torch.equal(vs[3].grad, torch.zeros(5))                         # block handed out, never used in the loss
    assert torch.equal(vs[4].grad, torch.full((7,), 4.0))                  # fallback path (+ the unused bank block's zeros)


def test_zero_grad_identity_keeps_the_parameter_in_the_graph():
    """x3d_coarse._ZeroGradFor: the Grid Pool conv biases cancel inside the batch-statistics BN; they stay graph inputs with an exactly-zero
    gradient so that SGD's weight decay / momentum treat them as in the reference"""
    import x3d_coarse as xc
    b2 = torch.randn(2, 5, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(5, requires_grad=True)
    out = xc._ZeroGradFor.apply(b2 * 1.0, bias)
    (out * 3.0).sum().backward()
    assert torch.equal(b2.grad, torch.full((2, 5), 3.0, dtype=torch.float64))
    assert bias.grad is not None and bias.grad.dtype == torch.float32 and torch.equal(bias.grad, torch.zeros(5))
