#!/usr/bin/env python3
"""Attribute the small ATen kernels (casts, fills, copies, adds) of one x3d_coarse fineFEAT train step to their call sites (GPU box)."""
import os
import sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import torch.optim as optim
from torch.profiler import profile, ProfilerActivity
import cfn_hip
from cfn_hip import dist as cdist
import train_coarse_fineFEAT as tc

dev = torch.device('cuda')
cfn_hip.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
net = tc.build_model(dev, pretrained=None)
net.train(True)
opt = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(B, 1, int(os.environ.get('FRAMES', '64')))))
x = x.view((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))
x, labels, masks, fm, meta = x.to(dev), labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
feat = {k: v.to(dev) for k, v in feat.items()}
for _ in range(2):
    tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)
    torch.cuda.synchronize()
evs = prof.events()
kern = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
print('device kernels / memcpys in the step:', len(kern))
kc = Counter(e.name[:70] for e in kern)
for k, v in kc.most_common(25):
    print('%5d  %s' % (v, k))
# ATen ops by python call site (second pass, CPU activity only)
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof2:
    tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)
    torch.cuda.synchronize()
names = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::_to_copy', 'aten::clone', 'aten::mul', 'aten::sum',
         'aten::div', 'aten::cat', 'aten::sub', 'aten::neg', 'aten::mul_', 'aten::div_', 'aten::mean', 'aten::sigmoid', 'aten::max')
cnt = Counter()
for ev in prof2.key_averages(group_by_stack_n=6):
    if ev.key in names:
        st = [x for x in (ev.stack or []) if '.py' in x and 'profiler' not in x and 'torch/' not in x][:3]
        cnt[(ev.key, ' <- '.join(x.split('/')[-1][:60] for x in st))] += ev.count
print('--- ATen ops by call site')
for (k, st), v in cnt.most_common(100):
    print('%5d %-16s %s' % (v, k, st))
# the ops the autograd engine / optimizer issue carry no python frame: group those by operand shapes instead
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof3:
    tc.train_step(net, red, opt, x, labels, masks, feat, fm, meta)
    torch.cuda.synchronize()
sc = Counter()
for ev in prof3.key_averages(group_by_input_shape=True):
    if ev.key in ('aten::add_', 'aten::add', 'aten::mul', 'aten::sum', 'aten::copy_', 'aten::_to_copy', 'aten::clone', 'aten::fill_', 'aten::zero_'):
        sc[(ev.key, str(ev.input_shapes)[:110])] += ev.count
print('--- ATen ops by operand shapes')
for (k, sh), v in sc.most_common(60):
    print('%5d %-16s %s' % (v, k, sh))
if len(sys.argv) > 2:          # full python stacks of one op, e.g. `glue_profile_coarse.py 2 aten::clone`
    print('--- full stacks of', sys.argv[2])
    for ev in prof2.key_averages(group_by_stack_n=14):
        if ev.key == sys.argv[2]:
            print(ev.count, [x.split('/')[-1][:50] for x in (ev.stack or [])])
