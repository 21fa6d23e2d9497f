#!/usr/bin/env python3
"""Host time to ENQUEUE one train step against its device time, per stream at the benchmark sizes: a host time close to the device time means the
step is launch bound or synchronises somewhere (round 5: the joint step did)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                              # noqa: E402
import torch.optim as optim               # noqa: E402
import train_fine                         # noqa: E402
import train_joint as tj                  # noqa: E402
import train_coarse_fineFEAT as tc        # noqa: E402
import x3d_fine                           # noqa: E402
from cfn_hip import dist as cdist         # noqa: E402

dev = torch.device('cuda:0')


def measure(name, step, n=6):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    host.sort()
    total.sort()
    print('%-28s host enqueue %.1f ms, step %.1f ms' % (name, host[len(host) // 2], total[len(total) // 2]), flush=True)


B = 8
net = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.5).to(dev).train(True)
opt = optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
x = torch.randn(B, 3, 256, 224, 224, device=dev)
lab = (torch.rand(B, 157, 2560, device=dev) < 0.05).float()
m = torch.ones(B, 2560, device=dev)
measure('fine 8 x 256 frames', lambda: train_fine.train_step(net, red, opt, x, lab, m))
red.close()
del net, opt, x
torch.cuda.empty_cache()
for T in (64, 256):
    cn = tc.build_model(dev, pretrained=None)
    copt = optim.SGD(tc.param_groups(cn, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
    xc, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(B, 1, T, seed=1)))
    xc = xc[:, 0].contiguous().to(dev)
    labels, masks, fm, meta = labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
    feat = {k: v.to(dev) for k, v in feat.items()}
    cn.train(True)
    cred = cdist.GradReducer(cn.parameters())
    measure('coarse 8 x %d frames' % T, lambda: tc.train_step(cn, cred, copt, xc, labels, masks, feat, fm, meta))
    cred.close()
    del cn, copt, xc
    torch.cuda.empty_cache()
fine_net, net2 = tj.build_models(dev)
jopt = optim.SGD(tj.param_groups(fine_net, net2, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
xj = torch.randn(B, 3, 128, 224, 224, device=dev)
tl = 640
labj = (torch.rand(B, 157, tl, device=dev) < 0.05).float()
mj = torch.ones(B, tl, device=dev)
jred = cdist.GradReducer(list(fine_net.parameters()) + list(net2.parameters()))
measure('joint 8 x 128 + 64 frames', lambda: tj.train_step(fine_net, net2, jred, jopt, xj, labj, mj))
jred.close()
