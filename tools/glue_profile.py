#!/usr/bin/env python3
"""Attribute the small ATen kernels (casts, fills, copies, adds) of one train step to their call sites (GPU box)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import torch.optim as optim
from torch.profiler import profile, ProfilerActivity
import cfn_hip
from cfn_hip import dist as cdist
import train_fine

dev = torch.device('cuda')
cfn_hip.load()
net = train_fine.build_model(dev, pretrained=None)
net.train(True)
opt = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(1, 3, T, 224, 224, device=dev)
labels = (torch.rand(1, 157, T * 10, device=dev) < 0.05).float()
masks = torch.ones(1, T * 10, device=dev)
for _ in range(2):
    train_fine.train_step(net, red, opt, x, labels, masks)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train_fine.train_step(net, red, opt, x, labels, masks)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=25, max_name_column_width=40))
names = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::to', 'aten::_to_copy', 'aten::clone',
         'aten::contiguous', 'aten::zeros', 'aten::mul', 'aten::sum', 'aten::div')
from collections import Counter
cnt = Counter()
for ev in prof.events():
    if ev.name in names:
        st = [s for s in ev.stack if '.py' in s and 'profiler' not in s][:3]
        cnt[(ev.name, ' <- '.join(s.split('/')[-1][:48] for s in st))] += 1
for (k, st), v in cnt.most_common(60):
    print('%5d %-18s %s' % (v, k, st))
