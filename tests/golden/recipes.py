"""Deterministic input recipes shared by gen_golden.py (which runs the real
reference in the build container) and by the parity tests (which regenerate
the same inputs on any box).  Everything uses numpy's legacy RandomState,
whose stream is frozen across numpy versions.

Nothing here comes from the reference: these are the synthetic input
distributions SURVEY.md §8c prescribes.
"""
import numpy as np


def unit_nonneg_rows(rs, n, e):
    """|randn| rows, L2-normalised: what a ReLU + l2_normalize head emits."""
    x = np.abs(rs.randn(n, e))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def clustered_embeddings(seed, p, k, e, sigma):
    """P class centres, K noisy members each, class-contiguous rows, unit
    non-negative rows."""
    r = np.random.RandomState(seed)
    c = np.abs(r.randn(p, e))
    x = np.abs(np.repeat(c, k, axis=0) + sigma * r.randn(p * k, e))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def triplet_rows(seed, t, e, kind):
    """[T,3E] = concat(a,p,n).  kind: 'unit' (post-ReLU unit rows) | 'randn'."""
    rs = np.random.RandomState(seed)
    if kind == "unit":
        parts = [unit_nonneg_rows(rs, t, e) for _ in range(3)]
    else:
        parts = [rs.randn(t, e).astype(np.float32) for _ in range(3)]
    return np.concatenate(parts, axis=1)


def triplet_edge_rows():
    """Hand-made rows on exactly representable values (E=4), margin 0.5:
    a==p, a==n, p==n, exact-zero hinge, just-inactive, known-answer 2.5."""
    a = np.array([[1, 0, 0, 0]] * 6, np.float32)
    p = np.array([[1, 0, 0, 0],      # a == p        -> pos 0
                  [0, 1, 0, 0],      # a == n below  -> neg 0, pos 2  => 2.5
                  [0, 1, 0, 0],      # p == n        -> pos == neg    => 0.5
                  [1, 0.5, 0, 0],    # pos .25, neg .75 -> exactly 0
                  [1, 0.5, 0, 0],    # pos .25, neg 1   -> -0.25 -> 0
                  [0, 1, 0, 0]], np.float32)
    n = np.array([[0, 0, 1, 0],
                  [1, 0, 0, 0],
                  [0, 1, 0, 0],
                  [1, 0.5, 0.5, 0.5],
                  [1, 0, 1, 0],
                  [1, 0, 0, 0]], np.float32)
    return np.concatenate([a, p, n], axis=1)


def siamese_pairs(seed, b, hi):
    """d ~ U(0,hi) [B,1]; y = first half 1, second half 0 (generator layout)."""
    rs = np.random.RandomState(seed)
    d = rs.uniform(0.0, hi, size=(b, 1)).astype(np.float32)
    y = np.zeros((b, 1), np.float32)
    y[: b // 2] = 1.0
    return y, d


TRIPLET_CASES = [  # (name, T, E, margin, kind, seed)
    ("t4_e8_unit", 4, 8, 0.5, "unit", 11),
    ("t4_e8_randn", 4, 8, 0.5, "randn", 12),
    ("t9_e256_unit", 9, 256, 0.3, "unit", 13),
    ("t9_e256_randn", 9, 256, 0.3, "randn", 14),
    ("t60_e256_unit", 60, 256, 0.5, "unit", 15),
    ("t60_e256_randn", 60, 256, 0.5, "randn", 16),
    ("t192_e256_unit", 192, 256, 0.5, "unit", 17),
    ("t192_e256_randn", 192, 256, 0.5, "randn", 18),
    ("t384_e512_unit", 384, 512, 0.5, "unit", 19),
]

SIAMESE_CASES = [  # (name, B, hi, seed)
    ("b8_sigmoid", 8, 1.0, 21),
    ("b8_l2", 8, float(np.sqrt(2.0)), 22),
    ("b256_sigmoid", 256, 1.0, 23),
    ("b256_l2", 256, float(np.sqrt(2.0)), 24),
]

PAIRWISE_CASES = [  # (name, N, E, seed, duplicate_row)
    ("n9_e256", 9, 256, 31, False),
    ("n60_e256", 60, 256, 32, True),
    ("n32_e256", 32, 256, 33, False),
    ("n128_e256", 128, 256, 34, True),
    ("n256_e512", 256, 512, 35, False),
    ("n512_e4096", 512, 4096, 36, False),
]

MINING_CASES = [  # (name, P, K, E, margin, sigma, seed)
    ("template_3x3", 3, 3, 256, 0.3, 0.5, 0),
    ("roadsigns_20x3", 20, 3, 256, 0.5, 0.25, 0),
    ("c1_8x4", 8, 4, 256, 0.5, 0.25, 0),
    ("c2_32x4", 32, 4, 256, 0.5, 0.25, 0),
    ("c5_64x4", 64, 4, 512, 0.5, 0.25, 0),
    ("c1_8x4_saturated", 8, 4, 256, 0.5, 0.5, 1),
    ("c1_8x4_fallback", 8, 4, 256, 0.5, 0.1, 2),
]
MINING_MODES = ["hardest", "semihard", "random_hard"]


def pairwise_input(n, e, seed, dup):
    rs = np.random.RandomState(seed)
    x = unit_nonneg_rows(rs, n, e)
    if dup:                       # sampling with replacement -> identical rows
        x[n // 2] = x[1]
    return x


KNN_CASES = [  # (name, classes, per_class, E, sigma, n_query, seed)
    ("c10_e64", 10, 20, 64, 0.35, 50, 41),
    ("c107_e256", 107, 6, 256, 0.3, 120, 42),
]


def knn_data(n_classes, per_class, e, sigma, n_query, seed):
    """Clustered unit embeddings: gallery [classes*per_class, e] with labels, and queries drawn the same way."""
    x = clustered_embeddings(seed, n_classes, per_class + 2, e, sigma).reshape(n_classes, per_class + 2, e)
    gallery = x[:, :per_class].reshape(-1, e)
    labels = np.repeat(np.arange(n_classes), per_class)
    rs = np.random.RandomState(seed + 1)
    pick = rs.randint(0, n_classes * 2, size=n_query)
    queries = x[:, per_class:].reshape(-1, e)[pick]
    return gallery.copy(), labels, queries.copy(), pick // 2
