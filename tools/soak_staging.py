#!/usr/bin/env python3
"""Soak of the staged input path (round 6): N coarse train steps fed by cfn_hip.staging from a loader that alternates pinned and pageable batches of two
different sizes (the slabs grow once, then must stay); device bytes, pinned slab bytes and host RSS are printed every 50 steps and must be flat.

    python tools/soak_staging.py [--steps 400] > profiles/r06_soak_staging.txt"""
import argparse
import os
import resource
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                         # noqa: E402
import torch.optim as optim          # noqa: E402
import train_coarse_fineFEAT as tc   # noqa: E402
from cfn_hip import dist as cdist    # noqa: E402
from cfn_hip.staging import HostStager, _map_tensors  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=400)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    net = tc.build_model(dev, pretrained=None).train(True)
    opt = optim.SGD(tc.param_groups(net, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
    red = cdist.GradReducer(net.parameters())
    small = next(iter(tc.SyntheticCoarse(2, 1, 16, seed=1)))
    big = next(iter(tc.SyntheticCoarse(3, 1, 32, seed=2)))
    variants = []
    for b in (small, big):
        x, labels, masks, feat, fm, meta, _, _ = b
        host = (x[:, 0].contiguous(), labels, masks, feat, fm, meta)
        variants += [host, _map_tensors(host, lambda t: t.pin_memory())]

    def loader():
        for i in range(a.steps):
            yield variants[i % len(variants)]
    st = HostStager(dev)
    print('# step   device MB   device peak MB   host RSS MB   slab bytes (device / pinned)   loss')
    for i, batch in enumerate(st.stage(loader())):
        cls, loc, _ = tc.train_step(net, red, opt, *batch)
        if i % 50 == 49 or i == a.steps - 1:
            torch.cuda.synchronize()
            slabs = (sum(s.dev.numel() for s in st.slots if s.dev is not None), sum(s.host.numel() for s in st.slots if s.host is not None))
            print('%6d %11.2f %16.2f %13.1f   %d / %d   %.5f' % (i + 1, torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.max_memory_allocated() / 2 ** 20,
                                                             resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, slabs[0], slabs[1], float(loc)), flush=True)
    print('# batches staged: %d, bytes staged: %d' % (st.batches, st.bytes_staged))
