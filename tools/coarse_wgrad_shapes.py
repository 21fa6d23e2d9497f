#!/usr/bin/env python3
"""Lists the (kernel entry, shape) of every weight-gradient / fusion call of one x3d_coarse train step (batch 8, T = 256) with its count:
which layers are behind pw_wgrad_kernel / fusion_gather in the coarse profile."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import torch.optim as optim
import cfn_hip
from cfn_hip import dist as cdist
import train_coarse_fineFEAT as tc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda')
cfn_hip.load()
net = tc.build_model(dev, pretrained=None)
net.train(True)
opt = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(8, 1, T)))
x = x.view((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))
x, labels, masks, fm, meta = x.to(dev), labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
feat = {k: v.to(dev) for k, v in feat.items()}
tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)
seen = collections.Counter()
orig = cfn_hip.call
def spy(name, *args):
    if 'bwd_weight' in name or 'fusion' in name or 'wgrad' in name:
        seen[(name, tuple(a for a in args if isinstance(a, int)))] += 1
    return orig(name, *args)
cfn_hip.call = spy
import cfn_hip.ops as ops_mod
ops_mod.call = spy
tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
