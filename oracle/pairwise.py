"""Oracle (test infrastructure): the all-pairs Euclidean matrix the reference
obtains from scikit-learn at /root/reference/embedding_net/datagenerators.py:219
(`pairwise_distances(all_embeddings)`, default metric 'euclidean').

scikit-learn is a third-party dependency (requirements.txt:4, unpinned; 1.7.2
is what the build container has and what the golden vectors were made with).
Its published algorithm for float32 input (sklearn/metrics/pairwise.py,
`_euclidean_distances` + `_euclidean_distances_upcast`):
    upcast to float64; d = -2 X Xᵀ + ‖x‖² + ‖y‖²; cast to float32;
    clamp at 0; set the diagonal to 0 (X is Y); sqrt.
Pinned by tests/golden/pairwise_distances.npz (real sklearn output).
"""
import numpy as np


def pairwise_sqdist(x):
    x64 = np.asarray(x, np.float32).astype(np.float64)
    nn = np.sum(x64 * x64, axis=1)
    d = -2.0 * (x64 @ x64.T)
    d += nn[:, None]
    d += nn[None, :]
    d = d.astype(np.float32)
    np.maximum(d, 0, out=d)
    np.fill_diagonal(d, 0)
    return d


def pairwise_distances(x):
    d = pairwise_sqdist(x)
    np.sqrt(d, out=d)
    return d
