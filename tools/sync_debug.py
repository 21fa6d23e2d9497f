#!/usr/bin/env python3
"""One train step of each stream (fine, coarse, joint) under torch.cuda.set_sync_debug_mode("warn"): every host synchronisation inside a step is
printed with the repository frames that caused it (round 5: a torch.tensor(list, device=...) in train_joint.joint_forward cost the joint step 5 ms)."""
import sys, os, warnings, traceback, torch
sys.path.insert(0, "coarse-fine-networks_amd"); sys.path.insert(0, ".")
import torch.optim as optim
import train_fine, train_joint as tj
import train_coarse_fineFEAT as tc
from cfn_hip import dist as cdist
import x3d_fine
dev = torch.device("cuda:0")
def show(message, category, filename, lineno, file=None, line=None):
    print("SYNCWARN", str(message)[:100])
    for l in traceback.format_stack(limit=12)[:-1]:
        if "/root/repo" in l or "repo/" in l: print("   ", l.strip().splitlines()[0][:160])
warnings.showwarning = show
warnings.simplefilter("always")
net = x3d_fine.generate_model("M", n_classes=157, task="loc", base_bn_splits=1, dropout=0.5).to(dev).train(True)
opt = optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
x = torch.randn(2, 3, 16, 224, 224, device=dev); lab = (torch.rand(2, 157, 160, device=dev) < 0.05).float(); m = torch.ones(2, 160, device=dev)
for _ in range(2): train_fine.train_step(net, red, opt, x, lab, m)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
print("=== fine step"); train_fine.train_step(net, red, opt, x, lab, m)
torch.cuda.set_sync_debug_mode("default"); torch.cuda.synchronize(); red.close()
cn = tc.build_model(dev, pretrained=None); copt = optim.SGD(tc.param_groups(cn, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
xc, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(2, 1, 16, seed=1)))
xc = xc[:, 0].contiguous().to(dev); labels, masks, fm, meta = labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev); feat = {k: v.to(dev) for k, v in feat.items()}
cn.train(True); cred = cdist.GradReducer(cn.parameters())
for _ in range(2): tc.train_step(cn, cred, copt, xc, labels, masks, feat, fm, meta)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
print("=== coarse step"); tc.train_step(cn, cred, copt, xc, labels, masks, feat, fm, meta)
torch.cuda.set_sync_debug_mode("default"); torch.cuda.synchronize(); cred.close()
fine_net, net2 = tj.build_models(dev)
jopt = optim.SGD(tj.param_groups(fine_net, net2, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
xj = torch.randn(2, 3, 32, 224, 224, device=dev); tl = 160
labj = (torch.rand(2, 157, tl, device=dev) < 0.05).float(); mj = torch.ones(2, tl, device=dev)
jred = cdist.GradReducer(list(fine_net.parameters()) + list(net2.parameters()))
for _ in range(2): tj.train_step(fine_net, net2, jred, jopt, xj, labj, mj)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
print("=== joint step"); tj.train_step(fine_net, net2, jred, jopt, xj, labj, mj)
torch.cuda.set_sync_debug_mode("default"); torch.cuda.synchronize(); jred.close()
# staged inputs (cfn_hip/staging.py): pageable AND pinned host batches through the copy stream; the consumer's thread must not synchronise either
from cfn_hip.staging import HostStager
stager = HostStager(dev)
host_pageable = (xc.cpu(), labels.cpu(), masks.cpu(), {k: v.cpu() for k, v in feat.items()}, fm.cpu(), meta.cpu())
host_pinned = tuple({k: v.pin_memory() for k, v in t.items()} if isinstance(t, dict) else t.pin_memory() for t in host_pageable)
cred = cdist.GradReducer(cn.parameters())
it = stager.stage([host_pageable, host_pinned, host_pageable, host_pinned])
tc.train_step(cn, cred, copt, *next(it))
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
print("=== coarse steps on staged batches (pinned, pageable, pinned)")
for batch in it: tc.train_step(cn, cred, copt, *batch)
torch.cuda.set_sync_debug_mode("default"); torch.cuda.synchronize(); cred.close()
print("done")
