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
