#!/usr/bin/env python3
"""Forward of the Grid Pool saliency convs (24 -> 24, 3x3x3, stride 2; x3d_coarse.py:362-366) at the metric's shapes: the split-bf16 kernel of
csrc/salconvb.hip (CFN_SAL_BF16=1, default) against the exact-fp32 MFMA kernel of csrc/salconv.hip (CFN_SAL_BF16=0), one setting per process
(the library reads its switches once); device time by HIP events, results compared with each other and, on a small case, with fp64 on the CPU.

    python tools/salb_bench.py [--batch 8] [--frames 256]"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
import torch                      # noqa: E402

DEV = 'cuda'
CASES = (('conv1 56->28', 1, 56, False), ('conv2 28->14', 2, 28, True))


def child(out, batch, frames):
    import cfn_hip
    from cfn_hip import ops
    import torch.nn.functional as F
    cfn_hip.load()
    g = torch.Generator().manual_seed(0)
    res = {}
    for name, tdiv, H, pro in CASES:
        T = frames // tdiv
        x = torch.randn(batch, 24, T, H, H, generator=g).to(DEV)
        w = (torch.randn(24, 24, 3, 3, 3, generator=g) * (2.0 / 648) ** 0.5).to(DEV)
        A = (1 + 0.2 * torch.randn(batch, 24, generator=g)).to(DEV) if pro else None
        B = (0.3 * torch.randn(batch, 24, generator=g)).to(DEV) if pro else None
        act = 1 if pro else 0
        fwd = lambda: ops.conv3d_dense(x, w, (3, 3, 3), (2, 2, 2), (1, 1, 1), A, B, act, True)
        with torch.no_grad():
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y, s, q = fwd()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            same = all(torch.equal(y, fwd()[0]) for _ in range(5))
            # fp64 reference on a slice: sample 0, the first 9 input frames
            xs = x[:1, :, :9].cpu().double()
            if pro:
                xs = torch.relu(xs * A[:1].cpu().double().view(1, 24, 1, 1, 1) + B[:1].cpu().double().view(1, 24, 1, 1, 1))
            ref = F.conv3d(xs, w.cpu().double(), stride=2, padding=1)[:, :, :4]       # output frames 0..3 see input frames -1..7 only
            err = float((y[:1, :, :4].cpu().double() - ref).abs().max() / ref.abs().max())
        by = 4.0 * batch * 24 * (T * H * H + ((T - 1) // 2 + 1) * (H // 2) ** 2)
        res[name] = dict(ms=ms, same=same, err=err, gbs=by / ms / 1e6, y=y.cpu(), s=s.cpu(), q=q.cpu())
    torch.save(res, out)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--child', default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.batch, a.frames)
        sys.exit(0)
    outs = {}
    for mode in ('0', '1'):
        f = tempfile.mktemp(suffix='.pt')
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', f, '--batch', str(a.batch), '--frames', str(a.frames)],
                              env=dict(os.environ, CFN_SAL_BF16=mode))
        outs[mode] = torch.load(f)
        os.remove(f)
    print('# saliency conv forward, %d clips x %d frames; algorithmic bytes = 4 B x (input + output elements)' % (a.batch, a.frames))
    for name, _, _, _ in CASES:
        r0, r1 = outs['0'][name], outs['1'][name]
        d = float((r1['y'].double() - r0['y'].double()).abs().max() / r0['y'].double().abs().max())
        ds = float((r1['s'] - r0['s']).abs().max() / r0['s'].abs().max())
        print('%s: exact fp32 MFMA %.3f ms (%.2f TB/s, err vs fp64 %.1e) | split bf16 %.3f ms (%.2f TB/s, err vs fp64 %.1e, bit-repeatable %s) | max |dy| / max |y| %.1e, statistics %.1e'
              % (name, r0['ms'], r0['gbs'] / 1e3, r0['err'], r1['ms'], r1['gbs'] / 1e3, r1['err'], r1['same'], d, ds))
