"""GPU: the drop-in entry points run end to end on the HIP path (synthetic Charades-shaped batches)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_train_fine_two_steps(tmp_path):
    import train_fine
    loaders = {'train': train_fine.SyntheticCharades(2, 3, frames=8, crop=64),
               'val': train_fine.SyntheticCharades(1, 1, frames=8, crop=64)}
    logs = []
    net = train_fine.run(batch_size=2, dataloaders=loaders, max_steps=2, pretrained=None, log=logs.append,
                         save_model=str(tmp_path / 'fine_'))
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert int(net.bn1.split_bn.num_batches_tracked) == 2


def test_train_coarse_step_and_val(tmp_path):
    import train_coarse_fineFEAT as tc
    loaders = {'train': tc.SyntheticCoarse(1, 1, frames=8, fine_len=12), 'val': tc.SyntheticCoarse(1, 1, frames=8, fine_len=12)}
    csvp = str(tmp_path / 'localize.csv')
    net = tc.run(batch_size=1, dataloaders=loaders, max_epochs=2, pretrained=None, csv_path=csvp, log=lambda *_: None,
                 save_model=str(tmp_path / 'coarse_'))
    assert all(torch.isfinite(p).all() for p in net.parameters())
    rows = open(csvp).read().strip().splitlines()
    assert len(rows) == 25 and len(rows[0].split(',')[2].split(' ')) == 157     # Charades_v1_localize row format


def test_extract_fine_features_roundtrip(tmp_path):
    import extract_fineFEAT as ex
    net = ex.build_tower(DEV, ckpt=None)
    g = torch.Generator().manual_seed(0)
    n = ex.extract(net, [('vidA', torch.randn(1, 3, 8, 224, 224, generator=g))], str(tmp_path))
    assert n == 1
    for k, c in (('layer1', 24), ('layer2', 48), ('layer3', 96), ('layer4', 192), ('conv5', 432)):
        f = torch.load(os.path.join(str(tmp_path), k, 'vidA'))
        assert f.shape == (1, c, 8, 7, 7) and f.dtype == torch.float32 and bool((f >= 0).all())


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


def test_grad_reducer_hooks_with_deferred_weight_casts():
    """the backward-overlapped bucket path (post-accumulate hooks -> flush of the lazily cast weight gradients -> pack ->
    all-reduce -> write back) on the GPU, one rank: gradients must equal those of a plain backward"""
    import torch.distributed as dist
    import train_fine
    from cfn_hip import dist as cdist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(0)
        net = train_fine.build_model(DEV, pretrained=None)
        net.train(False)   # running-statistics BN: well conditioned gradients, so the two passes can be compared tightly
        x = torch.randn(1, 3, 8, 64, 64, device=DEV)
        labels = (torch.rand(1, 157, 80, device=DEV) < 0.1).float()
        masks = torch.ones(1, 80, device=DEV)

        def grads(reducer):
            torch.manual_seed(7)                         # same dropout mask in both passes
            net.zero_grad(set_to_none=True)
            logits = net([x, masks[:, ::10]])
            cls_loss, loc_loss, _ = train_fine.detection_loss(logits, labels, masks, True)
            ((cls_loss + loc_loss) / 2).backward()
            reducer.finish()
            return [p.grad.detach().clone() for p in net.parameters() if p.grad is not None]

        state = {k: v.clone() for k, v in net.state_dict().items()}
        plain = grads(cdist.GradReducer(net.parameters()))
        net.load_state_dict(state)                       # same running statistics for the second pass
        hooked = grads(cdist.GradReducer(net.parameters(), bucket_bytes=1 << 18, force=True))
        assert len(plain) == len(hooked) > 200
        # a gradient read before its lazy cast would be garbage / non finite
        fa, fb = torch.cat([a.flatten() for a in plain]), torch.cat([b.flatten() for b in hooked])
        assert torch.isfinite(fb).all()
        assert float((fa - fb).norm() / fa.norm()) <= 1e-4
        for a, b in zip(plain, hooked):
            assert float((a - b).norm()) <= 1e-3 * float(a.norm()) + 1e-6
    finally:
        if created:
            dist.destroy_process_group()


def test_run_to_run_reproducibility():
    """Two identical forward+backward passes of the whole net on the same inputs.  All in-workgroup reductions have a fixed
    order and the cross-workgroup sums are fp64 accumulations of fp32 partials (exact for the magnitudes that occur), so the
    runs normally agree bit for bit (observed: 0.0 / 0.0); the thresholds only allow for a rare inexact fp64 sum, which
    train-mode BN at random init would amplify (a flipped last bit of a BN scale moved gradients by 1 % before the LDS
    float atomics were replaced by per-wave slots)."""
    import x3d_fine
    torch.manual_seed(0)
    net = x3d_fine.generate_model('M', n_classes=157, n_input_channels=3, task='loc', dropout=0.0, base_bn_splits=1).to(DEV)
    net.train(True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 8, 112, 112, generator=g).to(DEV)
    outs, grads = [], []
    for _ in range(2):
        for p in net.parameters():
            p.grad = None
        out = net([x, None])
        (out * out).mean().backward()
        outs.append(out.detach().clone())
        grads.append([p.grad.detach().clone() for p in net.parameters() if p.grad is not None])
    # forward statistics also go through fp64 atomics: a flipped last bit of a BN scale moves every logit slightly, and
    # train-mode BN amplifies it layer by layer (26 blocks)
    d_out = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    worst = 0.0   # per-parameter ||a - b|| / ||a||  (element-wise ratios are meaningless: the whole-net train-mode gradient is
    for a, b in zip(*grads):   # ill-conditioned in fp32)
        worst = max(worst, float((a - b).norm() / (a.norm() + 1e-30)))
    print('run-to-run: logits rel %.2e, worst gradient norm-rel %.2e' % (d_out, worst))
    assert d_out <= 1e-3, d_out
    assert worst <= 5e-2, worst


def _joint_nets(dropout=0.0):
    import train_joint
    from oracle import spec
    fine, coarse = train_joint.build_models(DEV, dropout=dropout)
    spec.fill_module_(fine)
    spec.fill_module_(coarse)
    coarse.rw6.dropout.p = 0.0
    return train_joint, fine, coarse


@pytest.mark.capture
def test_joint_two_stream_step_is_one_graph():
    """BASELINE configs[4]: fine tower -> feature dict -> coarse stream in ONE autograd graph.  (a) the logits equal running
    the two nets separately with the features detached in between; (b) the gradients that reach the Fine stream equal the
    chain rule applied by hand (coarse backward to the features, then the tower's backward from those feature gradients);
    (c) they reach the first layer of the Fine stream."""
    from oracle import spec
    tj, fine, coarse = _joint_nets()
    fine.train(True)
    coarse.train(True)
    clip = spec.rand_input(7, (2, 3, 16, 224, 224)).to(DEV)
    r = None

    def bn_state():
        return [b.clone() for m in (fine, coarse) for b in m.buffers()]

    def restore(state):
        for b, s in zip([b for m in (fine, coarse) for b in m.buffers()], state):
            b.copy_(s)

    state = bn_state()
    logits, _ = tj.joint_forward(fine, coarse, clip)
    assert logits.shape == (2, 157, 8)
    r = spec.rand_input(8, tuple(logits.shape)).to(DEV)
    (logits * r).sum().backward()
    g_joint = {n: p.grad.clone() for n, p in fine.named_parameters() if p.grad is not None}
    gc_joint = {n: p.grad.clone() for n, p in coarse.named_parameters() if p.grad is not None}
    assert 'conv1_s.weight' in g_joint and float(g_joint['conv1_s.weight'].abs().max()) > 0
    assert all(torch.isfinite(v).all() for v in g_joint.values())
    assert not any(n.startswith(('fc1.', 'fc2.')) for n in g_joint)      # the tower's own classifier is not in the graph

    # separately: tower -> detach -> coarse; then the chain rule by hand
    for m in (fine, coarse):
        m.zero_grad(set_to_none=True)
    restore(state)
    xc, s = tj.coarse_window(clip)
    feat, _ = fine([clip, None])
    leaves = {k: v.detach().requires_grad_(True) for k, v in feat.items()}
    meta = torch.tensor([[s, xc.shape[2], clip.shape[2], 1]] * 2, dtype=torch.int64, device=DEV)
    logits2 = coarse([xc, leaves, torch.ones(2, clip.shape[2], device=DEV), 0, meta])
    assert float((logits2 - logits).abs().max()) <= 1e-6 * float(logits.abs().max())
    (logits2 * r).sum().backward()
    keys = list(feat.keys())
    torch.autograd.backward([feat[k] for k in keys], [leaves[k].grad for k in keys])
    for n, p in fine.named_parameters():
        if p.grad is not None:
            a, b = p.grad, g_joint[n]
            assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-12, n
    for n, p in coarse.named_parameters():
        if p.grad is not None:
            assert float((p.grad - gc_joint[n]).norm()) <= 1e-5 * float(gc_joint[n].norm()) + 1e-12, n


def test_joint_logits_vs_oracle():
    """eval-mode joint forward against the CPU oracle (tower + coarse stream restated from the reference), 1e-3 on logits"""
    from oracle import spec, x3d_ref
    tj, fine, coarse = _joint_nets()
    fine.eval()
    coarse.eval()
    clip = spec.rand_input(9, (1, 3, 16, 224, 224))
    with torch.no_grad():
        logits, feat = tj.joint_forward(fine, coarse, clip.to(DEV))
    sd_f = spec.procedural_fill(spec.fine_keys('M', 157, 1))
    sd_c = spec.procedural_fill(spec.coarse_keys('M', 157, 1))
    with torch.no_grad():
        feat_o = x3d_ref.x3d_fine_forward(sd_f, clip, 'M', training=False, global_tower=True)
        xc, s = tj.coarse_window(clip)
        meta = torch.tensor([[s, 8, 16, 1]], dtype=torch.int64)
        ref = x3d_ref.x3d_coarse_forward(sd_c, [xc, feat_o, torch.ones(1, 16), 0, meta], 'M', training=False)
    for k in feat_o:
        assert float((feat[k].cpu() - feat_o[k]).abs().max()) <= 1e-4, k
    assert float((logits.cpu() - ref).abs().max()) <= 1e-3


def test_joint_run_two_steps(tmp_path):
    import train_joint
    loader = train_joint.SyntheticJoint(1, 2, fine_frames=16, coarse_frames=8)
    logs = []
    fine, coarse = train_joint.run(batch_size=1, dataloader=loader, max_steps=2, log=logs.append, save_model=str(tmp_path / 'j_'))
    assert len(logs) == 2
    assert all(torch.isfinite(p).all() for m in (fine, coarse) for p in m.parameters())
    assert int(fine.bn1.split_bn.num_batches_tracked) == 2 and int(coarse.bn1.split_bn.num_batches_tracked) == 2


def test_fine_validation_multicrop(tmp_path):
    """val batches with n = 3 crops per video: logits (b*n) are reduced by the max over crops against (b) labels
    (train_fine.py:183-207); the loop must run and log a finite mAP"""
    import train_fine

    class Crops(train_fine.SyntheticCharades):
        def __iter__(self):
            for x, labels, masks, names in super().__iter__():
                yield x.repeat(1, 3, 1, 1, 1, 1) + 0.01 * torch.arange(3).view(1, 3, 1, 1, 1, 1), labels, masks, names

    loaders = {'train': train_fine.SyntheticCharades(2, 1, frames=8, crop=64), 'val': Crops(1, 2, frames=8, crop=64)}
    logs = []
    train_fine.run(batch_size=2, dataloaders=loaders, max_epochs=4, pretrained=None, log=logs.append,
                   save_model=str(tmp_path / 'fine_'))
    val = [l for l in logs if ' val ' in l]
    assert len(val) == 1 and 'nan' not in val[0].lower()


@pytest.mark.capture
@pytest.mark.parametrize('stream', ['fine', 'coarse'])
def test_graphed_step_equals_eager_step(stream):
    """hipGraph capture of the whole train step (cfn_hip/graph.py): three replayed steps must leave the same parameters,
    BN statistics and losses as three eager steps from the same start (same kernels, same order: equal to fp32 rounding of
    the atomically accumulated statistics)."""
    import copy
    import torch.optim as optim
    import train_fine
    import train_coarse_fineFEAT as tc
    from cfn_hip import dist as cdist
    from cfn_hip.graph import GraphedStep
    torch.manual_seed(0)
    if stream == 'fine':
        net = train_fine.build_model(DEV, pretrained=None, dropout=0.0)
        batches = [(x.view((x.shape[0],) + tuple(x.shape[2:])).to(DEV), l.to(DEV), m.to(DEV))
                   for x, l, m, _ in train_fine.SyntheticCharades(2, 3, frames=8, crop=64)]
        mk = lambda n, o: (lambda x, l, m: train_fine.train_step(n, cdist.GradReducer(n.parameters()), o, x, l, m)[:2])
    else:
        net = tc.build_model(DEV, pretrained=None, dropout=0.0)
        net.rw6.dropout.p = 0.0
        batches = []
        for x, l, m, feat, fm, meta, _, _ in tc.SyntheticCoarse(1, 3, frames=8, fine_len=12):
            batches.append((x[:, 0].contiguous().to(DEV), l.to(DEV), m.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV),
                            meta.to(DEV)))
        mk = lambda n, o: (lambda x, l, m, f, fm, mt: tc.train_step(n, cdist.GradReducer(n.parameters()), o, x, l, m, f, fm, mt)[:2])
    net.train(True)
    net2 = copy.deepcopy(net)
    o1 = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
    o2 = optim.SGD(net2.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
    eager, graphed = mk(net, o1), GraphedStep(mk(net2, o2), optimizer=o2)     # call 1 eager, call 2 captures, call 3 replays
    for b in batches:
        le = [float(v) for v in eager(*b)]
        lg = [float(v) for v in graphed(*b)]
        assert all(abs(a - c) <= 1e-5 * max(abs(a), 1.0) for a, c in zip(le, lg)), (le, lg)
    assert len(graphed._graphs) == 1
    for (n1, p1), (_, p2) in zip(net.state_dict().items(), net2.state_dict().items()):
        d = float((p1.double() - p2.double()).abs().max())
        # (3e-5: atomically accumulated fp32 gradients -- the Grid Pool saliency convs -- differ run to run in the last bits, and three
        # SGD steps with momentum carry that to ~1e-5 of max |p|: observed 1.06e-5 on pool_1.conv1.weight once in ~10 runs)
        assert d <= 3e-5 * (float(p1.double().abs().max()) + 1e-3), (n1, d)
    # a changed learning rate is picked up (re-capture), not silently ignored
    for g in o2.param_groups:
        g['lr'] = 0.0
    before = [p.detach().clone() for p in net2.parameters()]
    graphed(*batches[0])
    assert len(graphed._graphs) == 1 and o2.param_groups[0]['lr'] == 0.0
    moved = max(float((a - b).abs().max()) for a, b in zip(before, [p.detach() for p in net2.parameters()]))
    assert moved == 0.0         # SGD at lr 0 leaves every parameter where it was: the re-captured graph holds the new rate


@pytest.mark.parametrize('launcher', ['torchrun', 'self', pytest.param('self-graph', marks=pytest.mark.capture)])
def test_bench_two_ranks_sharing_the_gpu(tmp_path, launcher):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), with the two
    ranks sharing the box's single GPU over gloo (RCCL needs a device per rank): parameter sync from rank 0, the bucketed
    all-reduce queued behind backward on a side stream, the global loss normaliser, max-over-ranks timing and the one JSON
    line are all exercised with the real kernels.  Small clips keep it short.  launcher = 'self': plain `python bench.py
    --gpus 2` (no torchrun), which must re-launch itself as two ranks."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CFN_DIST_BACKEND='gloo', CFN_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--frames', '16',
           '--batch', '2']
    if launcher.startswith('self'):
        cmd = [sys.executable] + cmd[cmd.index(os.path.join(ROOT, 'bench.py')):]
    if launcher == 'self-graph':      # two hipGraphs per rank around the eager all-reduce (GraphedDPStep); one more step so that replays are timed
        cmd += ['--graph']
        cmd[cmd.index('--steps') + 1] = '3' 
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc['n_gpus'] == 2 and doc['config']['dist']['world_size'] == 2 and doc['config']['parallelism'] == 'dp2'
    assert doc['value'] > 0 and all(v == v for v in doc['loss']['last_step_cls_loc'])
    assert 'cpu_baseline' not in doc          # N > 1: no CPU leg
    if launcher == 'self-graph':      # the line says what is (not) overlapped: the all-reduce runs after the replayed backward
        assert doc['config']['launch'].startswith('hipGraph replay') and 'not overlapped' in doc['config']['launch']
    else:
        assert doc['config']['launch'] == 'eager'


def test_joint_step_with_bf16_fine_tower():
    """BASELINE configs[4] ("fp16 MFMA pointwise"): the Fine stream of the joint step in bf16 (bf16 MFMA pointwise convs), its
    fp32 pooled features feeding the fp32 Coarse stream.  Logits against the all-fp32 joint forward at bf16 accuracy (eval
    mode), and a train step that reaches the first layer of the tower."""
    from oracle import spec
    import train_joint
    outs = {}
    for dt in (None, 'bf16'):
        fine, coarse = train_joint.build_models(DEV, dropout=0.0, fine_act_dtype=dt)
        spec.fill_module_(fine)
        spec.fill_module_(coarse)
        fine.eval()
        coarse.eval()
        clip = spec.rand_input(9, (1, 3, 16, 224, 224)).to(DEV)
        with torch.no_grad():
            outs[dt], _ = train_joint.joint_forward(fine, coarse, clip)
    err = float((outs['bf16'] - outs[None]).abs().max() / outs[None].abs().max())
    print('joint logits, bf16 tower vs fp32: rel-max %.2e' % err)
    assert err <= 2e-2
    fine.train(True)
    coarse.train(True)
    coarse.rw6.dropout.p = 0.0
    logits, _ = train_joint.joint_forward(fine, coarse, spec.rand_input(10, (2, 3, 16, 224, 224)).to(DEV))
    logits.square().mean().backward()
    g = fine.conv1_s.weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0


def test_forward_video_chunks_long_videos_like_the_reference():
    """train_coarse_fineFEAT.py:215-224: a video longer than 1005 frames is evaluated in 1000-frame chunks, `meta[:, 0]`
    (the clip's start inside the fine features) advanced by 1000 per chunk, logits concatenated in time.  A 1100-frame
    synthetic video (224 x 224: the fusion layers tile the 7 x 7 fine features by whole factors): forward_video == the two
    chunks run by hand, and differs from an (incorrect) evaluation that forgets to advance meta.  The 1000-frame chunk also
    takes layer 1 through the frame-range path of ops.pwconv (54 x 1000 x 112 x 112 x 4 B exceeds the 2 GiB descriptor)."""
    import train_coarse_fineFEAT as tc
    from oracle import spec
    net = tc.build_model(DEV, pretrained=None, dropout=0.0)
    spec.fill_module_(net)
    net.eval()
    g = torch.Generator().manual_seed(5)
    Tv, Tf = 1100, 160
    x = torch.randn(1, 3, Tv, 224, 224, generator=torch.Generator(device=DEV).manual_seed(6), device=DEV)
    feat = {k: torch.relu(torch.randn(1, c, Tf, 7, 7, generator=g)).to(DEV) for k, c in tc.FEAT_DEPTH.items()}
    fm = torch.ones(1, Tf, device=DEV)
    meta = torch.tensor([[10, Tv, 1500, 1]], dtype=torch.int64, device=DEV)
    with torch.no_grad():
        got = tc.forward_video(net, x, feat, fm, 0, meta)
        a = net([x[:, :, :1000].contiguous(), feat, fm, 0, meta])
        m2 = meta.clone(); m2[:, 0] += 1000
        b = net([x[:, :, 1000:].contiguous(), feat, fm, 0, m2])
        b_wrong = net([x[:, :, 1000:].contiguous(), feat, fm, 0, meta])
        short = tc.forward_video(net, x[:, :, :1004].contiguous(), feat, fm, 0, meta)      # < 1005 frames: one piece
        whole = net([x[:, :, :1004].contiguous(), feat, fm, 0, meta])
    want = torch.cat([a, b], dim=2)
    assert got.shape == want.shape and torch.equal(got, want)
    assert int(meta[0, 0]) == 10                                   # the caller's meta is not modified
    assert not torch.equal(b, b_wrong)                             # the start offset matters
    assert torch.equal(short, whole)


@pytest.mark.capture
@pytest.mark.parametrize('act', [None, 'fp16'])
def test_graphed_dp_step_equals_eager_step(act):
    """(act = 'fp16': ADVICE r5 -- forward_backward scales the loss, so the un-scaling has to be part of the captured optimizer graph:
    GraphedDPStep's post_reduce hook; without it the replayed steps are 4096 times too large.)
    GraphedDPStep (graph A: forward + loss + backward with the reducer suspended; eager bucketed all-reduce on the reducer's static
    flat buffers; graph B: optimizer step) against plain eager steps with the same reducer path, one rank with forced hooks: same
    losses, parameters and BN statistics after three steps."""
    import copy
    import torch.distributed as dist
    import torch.optim as optim
    import train_fine
    from cfn_hip import dist as cdist
    from cfn_hip.graph import GraphedDPStep
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29534')
        dist.init_process_group('nccl', rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(0)
        net = train_fine.build_model(DEV, pretrained=None, dropout=0.0, act_dtype=act)
        net.train(True)
        net2 = copy.deepcopy(net)
        if act == 'fp16':                      # one LossScaler per net (deepcopy would share nothing useful: drop the copy's, it is re-created)
            net2.__dict__.pop('_cfn_loss_scaler', None)
        batches = [(x.view((x.shape[0],) + tuple(x.shape[2:])).to(DEV), l.to(DEV), m.to(DEV))
                   for x, l, m, _ in train_fine.SyntheticCharades(2, 4, frames=8, crop=64)]
        o1 = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
        o2 = optim.SGD(net2.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
        r1 = cdist.GradReducer(net.parameters(), force=True)
        r2 = cdist.GradReducer(net2.parameters(), force=True)
        graphed = GraphedDPStep(lambda x, l, m, tot: train_fine.forward_backward(net2, x, l, m, mask_total=tot)[:2], r2, o2,
                                pre=lambda x, l, m: (cdist.global_mask_count(m),), post_reduce=lambda: train_fine.post_reduce(net2))
        for b in batches:
            le = [float(v) for v in train_fine.train_step(net, r1, o1, *b)[:2]]
            lg = [float(v) for v in graphed(*b)]
            assert all(abs(a - c) <= 1e-5 * max(abs(a), 1.0) for a, c in zip(le, lg)), (le, lg)
        assert len(graphed._graphs) == 1 and len(graphed._opt_graphs) == 1
        for (n1, p1), (_, p2) in zip(net.state_dict().items(), net2.state_dict().items()):
            d = float((p1.double() - p2.double()).abs().max())
            # (3e-5: atomically accumulated fp32 gradients -- the Grid Pool saliency convs -- differ run to run in the last bits, and three
            # SGD steps with momentum carry that to ~1e-5 of max |p|: observed 1.06e-5 on pool_1.conv1.weight once in ~10 runs)
            assert d <= (3e-5 if act is None else 2e-4) * (float(p1.double().abs().max()) + 1e-3), (n1, d)
        if act == 'fp16':
            sc = train_fine.loss_scaler(net2)
            assert float(sc.scale) == train_fine.LOSS_SCALE_FP16 and float(sc.found_inf) == 0.0
    finally:
        if created:
            dist.destroy_process_group()


def test_host_stager_delivers_batches_in_order_and_intact():
    """cfn_hip.staging.HostStager: nested batches (tuple / dict / strings) of pageable and pinned host tensors of several
    dtypes arrive on the device bit for bit and in order while earlier batches are still being consumed by (slow) kernels; slabs grow; a loader
    that raises surfaces in the consumer; an abandoned pass leaves the stager reusable; a staged train step equals the resident one."""
    from cfn_hip.staging import HostStager
    g = torch.Generator().manual_seed(0)

    def batch(i, n):
        b = (torch.randn(n, 3, 4, 8, 8, generator=g), (torch.rand(n, 5, generator=g) < 0.5).float(), {'a': torch.randn(n, 7, generator=g), 'm': torch.arange(n * 4).view(n, 4) + i},
             ['name%d' % i] * n, torch.randn(n, 6, generator=g).to(torch.bfloat16), torch.zeros(0))
        if i % 2:          # every other batch: pinned where possible
            b = (b[0].pin_memory(), b[1], {k: v.pin_memory() for k, v in b[2].items()}, b[3], b[4], b[5])
        return b
    batches = [batch(i, 2 + 3 * (i % 3) + (40 if i == 4 else 0)) for i in range(9)]      # batch 4 outgrows the slabs
    st = HostStager(DEV)
    busy = torch.randn(4096, 4096, device=DEV)
    got = []
    for b in st.stage(batches):
        for _ in range(3):
            busy = torch.tanh(busy @ busy * 1e-4)          # keep the compute stream behind the producer
        got.append((b[0] * 1.0, b[1].clone(), {k: v.clone() for k, v in b[2].items()}, b[3], b[4].clone(), b[5]))
    assert len(got) == len(batches) and st.batches == len(batches)
    for h, d in zip(batches, got):
        assert torch.equal(h[0], d[0].cpu()) and torch.equal(h[1], d[1].cpu()) and d[3] == h[3] and torch.equal(h[4], d[4].cpu())
        assert all(torch.equal(h[2][k], d[2][k].cpu()) for k in h[2]) and d[2]['m'].dtype == torch.int64
        assert d[0].is_cuda and d[5].numel() == 0

    def bad():
        yield batches[0]
        raise ValueError('loader failed')
    with pytest.raises(ValueError):
        for _ in st.stage(bad()):
            pass
    it = st.stage(batches)               # abandoned after one batch ...
    next(it)
    it.close()
    assert sum(1 for _ in st.stage(batches[:3])) == 3      # ... and the stager still works
    # a staged step == the resident step
    import copy
    import torch.optim as optim
    import train_fine
    from cfn_hip import dist as cdist
    torch.manual_seed(0)
    net = train_fine.build_model(DEV, pretrained=None, dropout=0.0).train(True)
    net2 = copy.deepcopy(net)
    data = [(x.view((x.shape[0],) + tuple(x.shape[2:])), l, m) for x, l, m, _ in train_fine.SyntheticCharades(2, 3, frames=8, crop=64)]
    o1, o2 = (optim.SGD(n.parameters(), lr=0.01, momentum=0.9) for n in (net, net2))
    r1, r2 = cdist.GradReducer(net.parameters()), cdist.GradReducer(net2.parameters())
    l1 = [float(train_fine.train_step(net, r1, o1, *[t.to(DEV) for t in b])[1]) for b in data]
    l2 = [float(train_fine.train_step(net2, r2, o2, *b)[1]) for b in st.stage(data)]
    assert l1 == l2


def test_dataloader_collate_and_staging_feed_the_coarse_run(tmp_path):
    """The host data path end to end (collate.py wired into a DataLoader; next-step 6: staging): a torch Dataset of RAGGED per-video
    samples shaped like charades_coarse_fineFEAT's (clips (1,3,T,224,224), labels (157,TL), 5 fine feature maps (C,T',7,7), meta, vid, duration)
    -> DataLoader(collate_fn=collate.coarse_collate, pin_memory=True, 2 workers) -> train_coarse_fineFEAT.run(dataloaders=...), whose loop stages the
    batches through cfn_hip.staging.  Two optimisation steps and one validation video: finite losses, parameters move, the CSV is written."""
    import numpy as np
    import torch.utils.data as tud
    import collate
    import train_coarse_fineFEAT as tc

    class Videos(tud.Dataset):
        def __init__(self, n, seed):
            self.n, self.seed = n, seed

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            r = np.random.RandomState(self.seed + i)
            tf = 20 + 4 * (i % 3)                                     # ragged fine-feature lengths
            tl = 160 - 10 * (i % 2)                                   # ragged label lengths (the clip window is fixed: 16 frames at stride 10)
            feat = {k: np.abs(r.randn(c, tf, 7, 7)).astype(np.float32) for k, c in tc.FEAT_DEPTH.items()}
            return (r.randn(1, 3, 16, 224, 224).astype(np.float32), (r.rand(157, tl) < 0.05).astype(np.float32), feat,
                    np.array([2 * (i % 3), 16, tf, 1], dtype=np.int64), 'vid%d' % i, 30.0 + i)
    loaders = {'train': tud.DataLoader(Videos(4, 0), batch_size=2, shuffle=False, num_workers=2, pin_memory=True, collate_fn=collate.coarse_collate),
               'val': tud.DataLoader(Videos(1, 100), batch_size=1, shuffle=False, num_workers=0, pin_memory=True, collate_fn=collate.coarse_collate)}
    logs = []
    csv_path = str(tmp_path / 'loc.csv')
    net = tc.run(max_epochs=2, batch_size=2, dataloaders=loaders, pretrained=None, save_model=str(tmp_path / 'm_'), csv_path=csv_path, log=logs.append)
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
    assert any('val Loc Loss' in l for l in logs) and all('nan' not in l.lower() for l in logs), logs
    assert os.path.getsize(csv_path) > 0
