#!/usr/bin/env python3
"""Run-to-run reproducibility scan (GPU box): K identical train-mode forward + backward passes of x3d_fine / x3d_coarse on the same
inputs; lists every output / gradient whose BITS differ between runs and by how much.  `--det 1` switches the C ABI's deterministic
mode on (cfn_deterministic) first.

    python tools/determinism_scan.py [--stream fine|coarse|both] [--runs 10] [--det 0|1] [--frames 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'coarse-fine-networks_amd')):
    sys.path.insert(0, p)
import torch                      # noqa: E402
import cfn_hip                    # noqa: E402

DEV = 'cuda'


def scan(name, net, inp, runs):
    import _init as spec
    net.train(True)
    ref = None
    worst = {}
    for r in range(runs):
        for p in net.parameters():
            p.grad = None
        y = net(inp)
        if r == 0:
            rr = spec.rand_input(777, tuple(y.shape)).to(DEV)
        (y * rr).sum().backward()
        torch.cuda.synchronize()
        cur = {'<logits>': y.detach().clone()}
        cur.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        if ref is None:
            ref = cur
            continue
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                d = float((v.double() - ref[k].double()).norm() / (ref[k].double().norm() + 1e-300))
                n, w = worst.get(k, (0, 0.0))
                worst[k] = (n + 1, max(w, d))
    print('%s: %d tensors compared over %d runs, %d differ in at least one run' % (name, len(ref), runs, len(worst)))
    for k, (n, w) in sorted(worst.items(), key=lambda kv: -kv[1][1])[:25]:
        print('   %-40s differs in %d of %d runs, worst norm-rel %.2e' % (k, n, runs - 1, w))
    return len(worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stream', default='both')
    ap.add_argument('--runs', type=int, default=10)
    ap.add_argument('--det', type=int, default=0)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--batch', type=int, default=2)
    a = ap.parse_args()
    cfn_hip.load()
    if a.det:
        cfn_hip.query('cfn_deterministic', 1)
    import _init as spec
    bad = 0
    if a.stream in ('fine', 'both'):
        import x3d_fine
        net = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
        spec.fill_module_(net)
        net.to(DEV)
        x = spec.rand_input(5, (a.batch, 3, a.frames, 224, 224)).to(DEV)
        bad += scan('x3d_fine', net, [x, None], a.runs)
    if a.stream in ('coarse', 'both'):
        import x3d_coarse
        depth = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}
        net = x3d_coarse.generate_model('M', n_classes=400, feat_depth=depth, task='loc', dropout=0.0, base_bn_splits=1,
                                        learnedMixing=True, isMixing=True, t_pool='grid')
        net.replace_logits(157)
        spec.fill_module_(net)
        net.to(DEV)
        net.rw6.dropout.p = 0.0
        B, T, Tf = a.batch, a.frames, 12
        x = spec.rand_input(6, (B, 3, T, 224, 224)).to(DEV)
        feat = {k: spec.rand_input(7 + i, (B, c, Tf, 7, 7), nonneg=True).to(DEV) for i, (k, c) in enumerate(depth.items())}
        fm = torch.ones(B, Tf, device=DEV)
        meta = torch.zeros(B, 4, dtype=torch.int64)
        for b in range(B):
            meta[b] = torch.tensor([b, T, Tf, 1])
        bad += scan('x3d_coarse', net, [x, feat, fm, 0, meta.to(DEV)], a.runs)
    print('TOTAL differing tensors:', bad)


if __name__ == '__main__':
    main()
