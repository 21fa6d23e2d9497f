"""N>1 path on CPU: two gloo processes run the product's gradient reducer and loss normaliser (cfn_hip.dist,
train_fine.detection_loss) and must reproduce the single-process, gathered-batch gradients -- i.e. the semantics
nn.DataParallel gives the reference."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import PKG

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv1d(6, 16, 1), nn.ReLU(), nn.Conv1d(16, 5, 1))


def _batch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 6, 8, generator=g)
    labels = (torch.rand(4, 5, 24, generator=g) < 0.3).float()
    masks = torch.ones(4, 24)
    masks[1, 15:] = 0
    masks[3, 6:] = 0
    return x, labels, masks


def _worker(rank, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from cfn_hip import dist as cdist
    import train_fine
    r, w, dev = cdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, WORLD) and dev.type == 'cpu'
    net = _model()
    reducer = cdist.GradReducer(net.parameters(), bucket_bytes=256)     # several buckets
    assert len(reducer.buckets) > 1
    x, labels, masks = _batch()
    sl = slice(rank * 2, rank * 2 + 2)
    for _ in range(2):                                                    # two steps: reducer state resets correctly
        net.zero_grad()
        cls, loc, _ = train_fine.detection_loss(net(x[sl]), labels[sl], masks[sl], align_corners=False)
        ((cls + loc) / 2).backward()
        reducer.finish()
    tot = cdist.global_mask_count(masks[sl])
    out[rank] = ([p.grad.clone() for p in net.parameters()], float(tot))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_match_gathered_batch():
    sys.path.insert(0, PKG)
    import train_fine
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    net = _model()
    x, labels, masks = _batch()
    cls, loc, _ = train_fine.detection_loss(net(x), labels, masks, align_corners=False)
    ((cls + loc) / 2).backward()
    ref = [p.grad for p in net.parameters()]
    for rank in range(WORLD):
        grads, tot = out[rank]
        assert abs(tot - float(masks.sum())) < 1e-6          # global loc-loss normaliser
        for g, r in zip(grads, ref):
            assert torch.allclose(g, r, rtol=1e-5, atol=1e-6), rank
    # both ranks hold identical (averaged) gradients
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b)


def test_param_groups_follow_the_reference_rule():
    """names containing 'rw' or 'mix' get 10x lr (train_coarse_fineFEAT.py:137-141): 1,204,447 of 4,532,327 params"""
    sys.path.insert(0, PKG)
    import train_coarse_fineFEAT as tc
    net = tc.build_model('cpu')
    groups = tc.param_groups(net, 0.02)
    n_base = sum(p.numel() for p in groups[0]['params'])
    n_rw = sum(p.numel() for p in groups[1]['params'])
    assert groups[1]['lr'] == pytest.approx(0.2)
    assert n_rw == 1204447 and n_base + n_rw == 4532327


# ---------------------------------------------------------------------------------------------------------------------
# replicas start identical: unseeded construction on every rank, then sync_module
# ---------------------------------------------------------------------------------------------------------------------
def _worker_sync(rank, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from cfn_hip import dist as cdist
    import train_fine
    cdist.init_from_env(backend='gloo')
    torch.manual_seed(100 + rank)                     # every rank draws DIFFERENT initial weights (as replace_logits does)
    net = nn.Sequential(nn.Conv1d(6, 16, 1), nn.BatchNorm1d(16), nn.ReLU(), nn.Conv1d(16, 5, 1))
    before = torch.cat([p.detach().flatten() for p in net.parameters()]).clone()
    cdist.sync_module(net)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    reducer = cdist.GradReducer(net.parameters(), bucket_bytes=128)
    x, labels, masks = _batch()
    sl = slice(rank * 2, rank * 2 + 2)
    for _ in range(3):
        cls, loc, _ = train_fine.detection_loss(net(x[sl]), labels[sl], masks[sl], align_corners=False)
        ((cls + loc) / 2).backward()
        reducer.finish()
        opt.step()
        opt.zero_grad(set_to_none=True)
    agree_all = cdist.all_agree(True, torch.device('cpu'))
    agree_one = cdist.all_agree(rank == 0, torch.device('cpu'))       # rank 1 has a short batch -> everybody skips
    gathered = cdist.gather_objects({'rank': rank, 'rows': [rank] * (rank + 1)})
    means = cdist.mean_over_ranks([float(rank), 10.0], torch.device('cpu'))
    out[rank] = (before, torch.cat([p.detach().flatten() for p in net.parameters()]),
                 net[1].running_mean.clone(), agree_all, agree_one, gathered, means, cdist.describe())
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_identical_after_sync_and_training():
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_sync, args=(port, out), nprocs=WORLD, join=True)
    assert not torch.equal(out[0][0], out[1][0])               # the ranks really started from different draws
    assert torch.equal(out[0][1], out[1][1])                   # ... and hold ONE model after sync + 3 averaged steps
    for rank in range(WORLD):
        _b, _p, _rm, agree_all, agree_one, gathered, means, desc = out[rank]
        assert agree_all is True and agree_one is False
        assert means == [0.5, 10.0]
        assert desc['world_size'] == WORLD and desc['backend'] == 'gloo'
    assert out[1][5] is None and [g['rank'] for g in out[0][5]] == [0, 1] and out[0][5][1]['rows'] == [1, 1]


def test_bucket_plan_on_the_real_parameter_sets():
    """GradReducer over the REAL x3d_fine / x3d_coarse parameter lists (CPU tensors, shapes only): every trainable
    parameter lands in exactly one bucket, buckets follow reverse registration order (the order backward produces
    gradients in), all but the last reach the size threshold."""
    sys.path.insert(0, PKG)
    from cfn_hip import dist as cdist
    import train_fine
    import train_coarse_fineFEAT as tc
    for net, total in ((train_fine.build_model('cpu'), 3296415),
                       (tc.build_model('cpu'), 4532327)):
        params = [p for p in net.parameters() if p.requires_grad]
        assert sum(p.numel() for p in params) == total
        red = cdist.GradReducer(params, bucket_bytes=4 << 20)
        flat = [p for b in red.buckets for p in b]
        assert len(flat) == len(params) and len(set(map(id, flat))) == len(params)
        assert [id(p) for p in flat] == [id(p) for p in reversed(params)]
        sizes = [sum(p.numel() * 4 for p in b) for b in red.buckets]
        assert all(s >= (4 << 20) for s in sizes[:-1]) and 3 <= len(sizes) <= 6, sizes
        assert red._pending == [len(b) for b in red.buckets]


def _worker_partial(rank, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from cfn_hip import dist as cdist
    cdist.init_from_env(backend='gloo')
    torch.manual_seed(0)
    used, unused = nn.Linear(4, 3), nn.Linear(4, 3)        # `unused` never contributes to the loss: no gradient at all
    params = list(used.parameters()) + list(unused.parameters())
    reducer = cdist.GradReducer(params, bucket_bytes=1 << 20)      # ONE bucket holding used and unused parameters
    x = torch.full((2, 4), float(rank + 1))
    res = []
    for _ in range(2):
        for p in params:
            p.grad = None
        used(x).sum().backward()
        reducer.finish()                                    # the half-filled bucket must still be reduced, every step
        res.append(used.weight.grad.clone())
    out[rank] = (res, unused.weight.grad)
    dist.barrier()
    dist.destroy_process_group()


def test_finish_reduces_buckets_with_gradientless_parameters():
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_partial, args=(port, out), nprocs=WORLD, join=True)
    for rank in range(WORLD):
        res, unused_grad = out[rank]
        assert unused_grad is None
        for g in res:                                       # mean over ranks of 2 * (rank + 1) = 3
            assert torch.allclose(g, torch.full((3, 4), 3.0))


def _worker_asymmetric(rank, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from cfn_hip import dist as cdist
    cdist.init_from_env(backend='gloo')
    torch.manual_seed(0)
    a, b, c = nn.Linear(4, 3), nn.Linear(4, 3), nn.Linear(4, 3)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    reducer = cdist.GradReducer(params, bucket_bytes=1)               # one parameter per bucket: 6 buckets
    assert len(reducer.buckets) == 6
    x = torch.full((2, 4), float(rank + 1))
    res = []
    for step in range(3):
        for p in params:
            p.grad = None
        # data-dependent branch (rw6's dropout path, multi-crop): `b` is used by rank 1 only, `c` by nobody
        y = a(x).sum() + (b(x).sum() if rank == 1 else 0.0)
        y.backward()
        reducer.finish()                                              # must not hang: same collectives on both ranks
        res.append((a.weight.grad.clone(), None if b.weight.grad is None else b.weight.grad.clone(), c.weight.grad))
    flat_ids = [id(f) for f in reducer._flat]
    out[rank] = (res, len(set(flat_ids)) == 6 and all(f is not None for f in reducer._flat))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_saw_different_gradients_issue_the_same_collectives():
    """a bucket in which THIS rank saw no gradient at all used to be skipped; if another rank saw one
    the ranks issued different numbers of all-reduces (a hang).  Now every bucket is reduced every step (zero-filled where
    nothing arrived) and a parameter without a local gradient gets the average iff somebody had one."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_asymmetric, args=(port, out), nprocs=WORLD, join=True)
    for rank in range(WORLD):
        res, static = out[rank]
        assert static                                                  # one static flat buffer per bucket, reused
        for ga, gb, gc in res:
            assert torch.allclose(ga, torch.full((3, 4), 3.0))         # mean of 2 * (rank + 1)
            assert gb is not None and torch.allclose(gb, torch.full((3, 4), 2.0))   # rank 1's 2 * 2, averaged with rank 0's zero
            assert gc is None                                          # nobody had one: stays None, as in the reference


def _worker_accum(rank, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from cfn_hip import dist as cdist
    import train_fine
    cdist.init_from_env(backend='gloo')
    x, labels, masks = _batch()
    res = {}
    for mode in ('plain', 'no_sync'):
        net = _model()
        reducer = cdist.GradReducer(net.parameters(), bucket_bytes=256)
        for _ in range(2):                                                # two updates: the state resets after an accumulated one
            net.zero_grad()
            for mb in range(2):                                           # two micro-batches per update (num_steps_per_update = 2)
                i = rank * 2 + mb
                cls, loc, _ = train_fine.detection_loss(net(x[i:i + 1]), labels[i:i + 1], masks[i:i + 1], align_corners=False)
                if mode == 'no_sync' and mb == 0:
                    with reducer.no_sync():
                        ((cls + loc) / 4).backward()
                else:
                    ((cls + loc) / 4).backward()
            reducer.finish()
        res[mode] = [p.grad.clone() for p in net.parameters()]
        reducer.close()
        assert not any(getattr(p, '_cfn_lazy_grad_ok', False) for p in net.parameters()) and not reducer._hooks
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_two_backwards_per_finish():
    """a second backward() before finish() used to be dropped (the first pass's averaged snapshot was
    copied over the accumulated p.grad).  Both spellings -- plain (every bucket is re-sent by finish()) and no_sync() -- must
    give the mean over ranks of the per-rank accumulated gradients; the reference keeps the knob (train_fine.py:65)."""
    sys.path.insert(0, PKG)
    import train_fine
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_accum, args=(port, out), nprocs=WORLD, join=True)
    x, labels, masks = _batch()
    net = _model()
    for mb in range(2):      # the gathered micro-batch = what DataParallel would have seen: rank 0's and rank 1's sample mb
        i = [mb, 2 + mb]
        cls, loc, _ = train_fine.detection_loss(net(x[i]), labels[i], masks[i], align_corners=False)
        ((cls + loc) / 4).backward()
    ref = [p.grad for p in net.parameters()]
    for rank in range(WORLD):
        for mode in ('plain', 'no_sync'):
            for g, r in zip(out[rank][mode], ref):
                assert torch.allclose(g, r, rtol=1e-5, atol=1e-7), mode


def _worker_begin_pass(rank, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from cfn_hip import dist as cdist
    import train_fine
    cdist.init_from_env(backend='gloo')
    x, labels, masks = _batch()
    net = _model()
    other = nn.Parameter(torch.ones(3))                                   # not one of the reducer's parameters
    reducer = cdist.GradReducer(net.parameters(), bucket_bytes=256)
    # a backward that raises drops the engine's end-of-pass callback: the reducer must not stay "inside a pass"
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')
    try:
        reducer.begin_pass()
        (Boom.apply(net(x[:1])).sum() + net(x[:1]).sum()).backward()
    except RuntimeError:
        pass
    net.zero_grad()
    reducer.finish()                                                      # (both ranks: one collective per bucket, state back to idle)
    assert not reducer._in_pass and reducer._begun == 0 and reducer._passes == 0
    for _ in range(2):
        net.zero_grad()
        for mb in range(2):
            i = rank * 2 + mb
            reducer.begin_pass()
            if rank == 1 and mb == 0:                                     # this rank's first pass yields NO gradient for any reducer parameter:
                (other * 2.0).sum().backward()                            # no hook fires, no engine callback counts the pass here
            else:
                cls, loc, _ = train_fine.detection_loss(net(x[i:i + 1]), labels[i:i + 1], masks[i:i + 1], align_corners=False, local_norm=True)
                ((cls + loc) / 4).backward()
        reducer.finish()
    out[rank] = [p.grad.clone() for p in net.parameters()]
    dist.barrier()
    dist.destroy_process_group()


def test_explicit_pass_count_is_rank_independent():
    """the number of backward passes since the last finish() decides whether every bucket is sent again; counted from
    autograd-engine callbacks it differs between ranks when one rank's pass produces no gradient for any reducer parameter (rank 0 would
    re-send, rank 1 would not: mismatched collectives).  `begin_pass()` counts calls instead.  Also: a backward that raised leaves no state."""
    sys.path.insert(0, PKG)
    import train_fine
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_begin_pass, args=(port, out), nprocs=WORLD, join=True)
    x, labels, masks = _batch()
    net = _model()
    # rank 0 accumulated samples 0 and 1, rank 1 only sample 3; the reducer averages the per-rank sums over the 2 ranks (local_norm: rank 1 never
    # enters detection_loss in its first pass, so the loss must not hold a collective of its own)
    for i in (0, 1, 3):
        cls, loc, _ = train_fine.detection_loss(net(x[i:i + 1]), labels[i:i + 1], masks[i:i + 1], align_corners=False, local_norm=True)
        ((cls + loc) / 4).backward()
    ref = [p.grad / WORLD for p in net.parameters()]
    for rank in range(WORLD):
        for g, r in zip(out[rank], ref):
            assert torch.allclose(g, r, rtol=1e-5, atol=1e-7)
