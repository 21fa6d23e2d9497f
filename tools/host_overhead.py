#!/usr/bin/env python3
"""Host enqueue time vs device time of one train step (GPU box): is the step launch bound?"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch
import torch.optim as optim
import cfn_hip
from cfn_hip import dist as cdist
import train_fine

dev = torch.device('cuda')
cfn_hip.load()
net = train_fine.build_model(dev, pretrained=None)
net.train(True)
opt = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
red = cdist.GradReducer(net.parameters())
for T in (16, 256):
    x = torch.randn(1, 3, T, 224, 224, device=dev)
    labels = (torch.rand(1, 157, T * 10, device=dev) < 0.05).float()
    masks = torch.ones(1, T * 10, device=dev)
    for _ in range(3):
        train_fine.train_step(net, red, opt, x, labels, masks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        train_fine.train_step(net, red, opt, x, labels, masks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('T=%d: host enqueue %.2f ms/step, wall %.2f ms/step' % (T, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
