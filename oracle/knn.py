"""Oracle (test infrastructure): brute-force k-NN on embeddings, the arithmetic scikit-learn's
KNeighborsClassifier (reference models.py:15; third-party, unpinned, 1.7.2 here) performs for
algorithm='brute', metric Euclidean, uniform weights: euclidean_distances(Q, X) with f64 accumulation
for f32 input, the k smallest per row in ascending order, majority vote with ties to the smallest class.
Pinned by tests/golden/knn.npz (real sklearn output)."""
import numpy as np


def cross_distances(q, x):
    q64, x64 = np.asarray(q, np.float32).astype(np.float64), np.asarray(x, np.float32).astype(np.float64)
    d = -2.0 * (q64 @ x64.T) + (q64 * q64).sum(1)[:, None] + (x64 * x64).sum(1)[None, :]
    d = np.maximum(d.astype(np.float32), 0)
    return np.sqrt(d)


def kneighbors(q, x, k):
    d = cross_distances(q, x)
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(d, idx, 1), idx


def predict(q, x, labels, k):
    """labels: integer class ids [n]."""
    _, idx = kneighbors(q, x, k)
    votes = np.asarray(labels)[idx]
    return np.array([np.bincount(v).argmax() for v in votes])
