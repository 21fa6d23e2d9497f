import os, sys
sys.path.insert(0, '/root/repo/coarse-fine-networks_amd')
import torch
from cfn_hip import ops
NB=4; T=256
for c,H in ((216,14),(432,7),(54,56)):
    x = torch.randn(NB, c, T, H, H, device='cuda').requires_grad_(True)
    w = (torch.randn(c, 1, 3, 3, 3, device='cuda') * 0.2).requires_grad_(True)
    A = (torch.rand(NB, c, device='cuda') + 0.5).requires_grad_(True)
    B = (torch.randn(NB, c, device='cuda') * 0.1).requires_grad_(True)
    y, sm, sq = ops.dwconv3d(x, w, A, B, 1, 1, True)
    gy, gs, gq = torch.randn_like(y), torch.randn_like(sm) * 0.01, torch.randn_like(sq) * 0.001
    for _ in range(3):
        torch.autograd.grad((y, sm, sq), (x, w, A, B), (gy, gs, gq), retain_graph=True)
torch.cuda.synchronize()
