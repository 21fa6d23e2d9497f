import os, sys
sys.path.insert(0, '/root/repo/coarse-fine-networks_amd')
import torch, cfn_hip
from cfn_hip import ops
cfn_hip.load()
DEV = 'cuda'
# correctness vs fp64 on odd shapes, then timing is done by microbench with CFN_PWR=0/1
for (N, ci, co, T, H) in [(2, 96, 216, 2, 14), (1, 96, 216, 5, 14), (2, 48, 160, 3, 6), (1, 80, 250, 3, 10), (2, 96, 216, 1, 14), (1, 64, 129, 2, 4)]:
    for act in (0, 2):
        torch.manual_seed(1)
        x = torch.randn(N, ci, T, H, H, device=DEV); w = torch.randn(co, ci, 1, 1, 1, device=DEV) * 0.1
        A = torch.rand(N, ci, device=DEV) + 0.5; B = torch.randn(N, ci, device=DEV) * 0.1
        y, s, q = ops.pwconv(x, w, A, B, act, 1, True)
        z = x.double() * A.double().view(N, ci, 1, 1, 1) + B.double().view(N, ci, 1, 1, 1)
        if act == 2: z = z * torch.sigmoid(z)
        yr = torch.einsum('nkthw,mk->nmthw', z, w.double().view(co, ci))
        e = float((y.double() - yr).abs().max() / yr.abs().max())
        es = float((s - yr.sum((2, 3, 4))).abs().max() / yr.sum((2, 3, 4)).abs().max())
        eq = float((q - (yr * yr).sum((2, 3, 4))).abs().max() / (yr * yr).sum((2, 3, 4)).abs().max())
        print('N=%d %d->%d T=%d H=%d act=%d: y %.2e  sum %.2e  sumsq %.2e' % (N, ci, co, T, H, act, e, es, eq))
