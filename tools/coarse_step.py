#!/usr/bin/env python3
"""Time the Coarse-Fine (fineFEAT fusion) train step of x3d_coarse on a synthetic Charades-shaped batch (GPU box).

    python tools/coarse_step.py [batch] [steps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import torch.optim as optim
import cfn_hip
from cfn_hip import dist as cdist
import train_coarse_fineFEAT as tc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda')
cfn_hip.load()
net = tc.build_model(dev, pretrained=None)
net.train(True)
opt = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(B, 1, 64)))
x = x.view((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))      # (B,1,3,T,H,W) -> (B,3,T,H,W) as in run()
x, labels, masks, fm, meta = x.to(dev), labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
feat = {k: v.to(dev) for k, v in feat.items()}


def step():
    return tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('x3d_coarse fineFEAT train step, batch %d x 64 frames: host %.2f ms, wall %.2f ms/step = %.1f clips/s'
      % (B, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3, B * steps / (t2 - t0)))
