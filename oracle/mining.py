"""Oracle (test infrastructure): online triplet mining restated on indices.

Follows /root/reference/embedding_net/datagenerators.py:
  hardest_negative :188-190, random_hard_negative :192-194,
  semihard_negative :196-199, get_batch_triplets_mining :219-250
(the image loading / predict part :202-217 is outside the hot path: the
mining consumes a class-contiguous [P*K, E] embedding block).
Pinned by tests/golden/mining.npz.  `batch_hard` is NOT in the reference
(parity unpinned, see oracle/__init__.py).
"""
import numpy as np

from .pairwise import pairwise_distances

MODES = ("semihard", "hardest", "random_hard")


def sample_batch(n_samples_per_class, p, k, rng=None):
    """:202-205 — P classes without replacement, K sample indices per class WITH
    replacement.  n_samples_per_class: list of class sizes (class id = position).
    Returns (class_ids [P], sample_idx [P,K]); consumes the legacy np.random
    stream exactly like the reference."""
    rng = rng or np.random
    cls = rng.choice(len(n_samples_per_class), size=p, replace=False)
    idx = [rng.choice(n_samples_per_class[c], size=k, replace=True) for c in cls]
    return cls, np.asarray(idx)


def candidate_mask(loss_values, margin, mode):
    """Which negatives each selection rule may return (all of them for the two
    random rules, the arg-max for 'hardest')."""
    lv = np.asarray(loss_values)
    if mode == "hardest":                       # :189-190
        m = np.zeros(lv.shape, bool)
        k = int(np.argmax(lv))                  # first max on ties
        m[k] = lv[k] > 0
        return m
    if mode == "random_hard":                   # :193
        return lv > 0
    if mode == "semihard":                      # :197-198 (strict both sides)
        return np.logical_and(lv < margin, lv > 0)
    raise KeyError(mode)


def mine_triplets(dist, p, k, margin, mode, rng=None):
    """dist: [N,N] float32 non-squared distances, N = p*k, rows class-contiguous.

    Returns dict(triplets [T,3] int32, loss_values [pairs,N-k] f32,
    candidates [pairs,N-k] bool, selected [pairs] int32 (-1 = none),
    fallback bool).  Pair order = class by class, (i<j) lexicographic — the
    order itertools.combinations yields at :231.
    """
    rng = rng or np.random
    n = p * k
    dist = np.asarray(dist)
    trip, losses, cands, sel = [], [], [], []
    for c in range(p):
        lo, hi = c * k, (c + 1) * k
        neg = np.concatenate([np.arange(0, lo), np.arange(hi, n)])     # :228-230
        for i in range(lo, hi):
            for j in range(i + 1, hi):
                lv = dist[i, j] - dist[i, neg] + margin               # :235
                m = candidate_mask(lv, margin, mode)
                losses.append(lv)
                cands.append(m)
                idx = np.where(m)[0]
                if len(idx) == 0:
                    sel.append(-1)
                    continue
                pick = idx[0] if mode == "hardest" else rng.choice(idx)
                sel.append(int(pick))
                trip.append((i, j, int(neg[pick])))                   # :240-243
    fallback = len(trip) == 0
    if fallback:                                                      # :246-250
        trip.append((n - 2, n - 1, 0))
    return dict(triplets=np.asarray(trip, np.int32).reshape(-1, 3),
                loss_values=np.asarray(losses, np.float32),
                candidates=np.asarray(cands, bool),
                selected=np.asarray(sel, np.int32),
                fallback=fallback)


def mine_from_embeddings(emb, p, k, margin, mode, rng=None):
    """:219 + :225-250 in one call."""
    return mine_triplets(pairwise_distances(emb), p, k, margin, mode, rng)


def batch_hard(dist, p, k):
    """Hermans et al. batch-hard selection (build-defined; README.md:112 cites
    the paper, the code never implements it).  One triplet per anchor:
    farthest same-class sample, closest other-class sample; first index wins
    ties.  Returns [N,3] int32."""
    n = p * k
    dist = np.asarray(dist)
    out = np.zeros((n, 3), np.int32)
    for a in range(n):
        lo = (a // k) * k
        hi = lo + k
        same = np.arange(lo, hi)
        same = same[same != a]
        other = np.concatenate([np.arange(0, lo), np.arange(hi, n)])
        out[a] = (a, same[np.argmax(dist[a, same])], other[np.argmin(dist[a, other])])
    return out
