"""Per-class average precision accumulator with the reference's ``APMeter`` surface
(apmeter.py:22-136: ``reset`` / ``add(output, target, weight=None)`` / ``value``).

Host-side metric: scores are ranked per class with a stable descending sort and AP is the
mean of precision@rank over the positive ranks -- the same number the reference's torch loop produces
(pinned in tests/golden/loss_ap.npz).  Storage is a list of numpy blocks concatenated lazily instead of a
manually grown torch storage."""
import numpy as np
import torch


class APMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self._scores, self._targets, self._weights = [], [], []

    def add(self, output, target, weight=None):
        output = output.detach().cpu().numpy() if torch.is_tensor(output) else np.asarray(output)
        target = target.detach().cpu().numpy() if torch.is_tensor(target) else np.asarray(target)
        if output.ndim == 1:
            output = output.reshape(-1, 1)
        if target.ndim == 1:
            target = target.reshape(-1, 1)
        assert output.ndim == 2 and target.ndim == 2, 'wrong size (should be 1D or 2D with one column per class)'
        assert np.array_equal(target * target, target), 'targets should be binary (0 or 1)'
        if self._scores:
            assert target.shape[1] == self._targets[0].shape[1], \
                'dimensions for output should match previously added examples.'
        if weight is not None:
            weight = (weight.detach().cpu().numpy() if torch.is_tensor(weight) else np.asarray(weight)).reshape(-1)
            assert weight.shape[0] == target.shape[0], 'Weight dimension 1 should be the same as that of target'
            assert weight.min() >= 0, 'Weight should be non-negative only'
            self._weights.append(weight.astype(np.float32))
        self._scores.append(output.astype(np.float32))
        self._targets.append(target.astype(np.int64))

    def value(self):
        """(K,) float32 tensor of per-class AP; 0 when nothing was added (as the reference)."""
        if not self._scores:
            return 0
        scores, targets = np.concatenate(self._scores), np.concatenate(self._targets)
        weights = np.concatenate(self._weights) if self._weights else None
        n, k = scores.shape
        ap = np.zeros(k, dtype=np.float32)
        ranks = np.arange(1, n + 1, dtype=np.float32)
        for j in range(k):
            order = np.argsort(-scores[:, j], kind='stable')
            truth = targets[order, j].astype(np.float32)
            if weights is not None:
                w = weights[order]
                tp, rg = np.cumsum(truth * w, dtype=np.float32), np.cumsum(w, dtype=np.float32)
            else:
                tp, rg = np.cumsum(truth, dtype=np.float32), ranks
            prec = tp / rg
            ap[j] = prec[truth > 0].sum() / max(truth.sum(), 1)
        return torch.from_numpy(ap)
