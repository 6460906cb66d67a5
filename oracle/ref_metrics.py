"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the evaluation metrics the reference computes over the logits
(ACT/ops/utils.py:35-88), pinned against vectors produced by the reference's own functions
(tests/golden/g9_metrics.npz)."""
import numpy as np


def accuracy(output, target, topk=(1,)):
    """utils.py:35-49: percent of rows whose target is among the k largest logits."""
    order = np.argsort(-output, axis=1, kind="stable")
    return [100.0 * float(np.mean([target[i] in order[i, :k] for i in range(len(target))])) for k in topk]


def cal_map(output, labels):
    """utils.py:68-88 with get_multi_hot(..., assumes_starts_zero=False) (:51-66)."""
    y = labels.copy()
    nxt = 0
    for v in np.unique(y):
        if v >= 0:
            y[y == v] = nxt
            nxt += 1
    n, c = output.shape
    gt = np.zeros((n, c + 1), dtype=np.float32)
    for j in range(y.shape[1]):
        gt[np.arange(n), y[:, j]] = 1
    gt = gt[:, :c]
    e = np.exp(output - output.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    ap = np.zeros(c, dtype=np.float32)
    for k in range(c):
        order = np.argsort(-probs[:, k], kind="stable")
        truth = gt[order, k]
        prec = np.cumsum(truth) / np.arange(1, n + 1, dtype=np.float32)
        ap[k] = prec[truth > 0].sum() / max(float(truth.sum()), 1.0)
    return float(ap.mean() * 100), ap * 100
