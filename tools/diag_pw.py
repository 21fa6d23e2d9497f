import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'coarse-fine-networks_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch, collections
import ab_pw
for N, ci, co, T, H in [(2, 108, 48, 4, 28), (2, 216, 96, 2, 14)]:
    args = ab_pw.setup(N, ci, co, T, H)
    ref, gyy = ab_pw.once(*args)
    Q = T * H * H
    seen = 0
    for it in range(200):
        out, _ = ab_pw.once(*args, gyy=gyy)
        d = (out[1] != ref[1]).reshape(N, ci, Q)
        if d.any():
            idx = d.nonzero()
            seen += 1
            if seen <= 4:
                n_, c_, q_ = idx[:, 0], idx[:, 1], idx[:, 2]
                print('shape %d->%d run %d: %d bad elements; samples %s; channels %s; ch%%32 %s; tiles(q//32) %s; q%%32 %s' % (
                    ci, co, it, len(idx), sorted(set(n_.tolist())), sorted(set(c_.tolist()))[:12], sorted(set((c_ % 32).tolist()))[:40],
                    sorted(set((q_ // 32).tolist()))[:12], sorted(set((q_ % 32).tolist()))))
                vals = (out[1] - ref[1]).reshape(N, ci, Q)[n_[0], c_[0], (q_[0] // 32) * 32:(q_[0] // 32) * 32 + 32]
                print('   diff row:', [round(float(v), 4) for v in vals])
    print('shape %d->%d: %d / 200 runs differ; ntiles %d' % (ci, co, seen, (Q + 31) // 32))
