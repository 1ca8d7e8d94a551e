"""CPU: pin the oracle against the golden vectors produced by the real
reference (tests/golden/gen_golden.py).  No GPU, no product code."""
import numpy as np
import pytest

import recipes as R
from oracle import losses, mining, pairwise


@pytest.mark.parametrize("case", R.TRIPLET_CASES, ids=lambda c: c[0])
def test_triplet_loss(golden, case):
    name, t, e, m, kind, seed = case
    g = golden("triplet_loss")
    y = R.triplet_rows(seed, t, e, kind)
    if f"{name}/y_pred" in g:
        assert np.array_equal(g[f"{name}/y_pred"], y), "recipe drifted from the stored input"
    out = losses.triplet_loss(m)(None, y)
    assert out.shape == (t,)
    np.testing.assert_allclose(out, g[f"{name}/loss"], rtol=1e-12, atol=1e-12)


def test_triplet_loss_edges(golden):
    g = golden("triplet_loss")
    out = losses.triplet_loss(0.5)(None, g["edge/y_pred"])
    assert np.array_equal(out, g["edge/loss"])
    assert np.array_equal(out, [0.0, 2.5, 0.5, 0.0, 0.0, 2.5])   # known answers, SURVEY §3.3


def test_triplet_grad_matches_finite_difference():
    y = R.triplet_rows(5, 6, 8, "randn").astype(np.float64)
    up = np.linspace(0.5, 1.5, 6)
    g = losses.triplet_loss_grad(0.5, y, up)
    f = lambda z: float(np.sum(up * losses.triplet_loss(0.5)(None, z)))
    num = np.zeros_like(y)
    for idx in np.ndindex(*y.shape):
        d = np.zeros_like(y)
        d[idx] = 1e-6
        num[idx] = (f(y + d) - f(y - d)) / 2e-6
    np.testing.assert_allclose(g, num, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", [c[0] for c in R.SIAMESE_CASES] + ["edge"])
def test_contrastive_and_accuracy(golden, name):
    g = golden("siamese_losses")
    y, d = g[f"{name}/y_true"], g[f"{name}/y_pred"]
    np.testing.assert_allclose(losses.contrastive_loss(y, d), g[f"{name}/contrastive"], rtol=1e-12)
    assert losses.accuracy(y, d) == g[f"{name}/accuracy"]


@pytest.mark.parametrize("case", R.PAIRWISE_CASES, ids=lambda c: c[0])
def test_pairwise(golden, case):
    name, n, e, seed, dup = case
    g = golden("pairwise_distances")
    x = R.pairwise_input(n, e, seed, dup)
    if f"{name}/X" in g:
        assert np.array_equal(g[f"{name}/X"], x)
    ref = g[f"{name}/D"]
    d = pairwise.pairwise_distances(x)
    assert d.dtype == np.float32
    assert np.all(np.diag(d) == 0) and np.array_equal(d, d.T) == np.array_equal(ref, ref.T)
    # same algorithm, only the f64 BLAS summation order may differ -> <= 1 f32 ulp before sqrt
    np.testing.assert_allclose(d * d, ref * ref, rtol=0, atol=3e-7)
    if dup:
        assert d[1, n // 2] <= 1e-3 and ref[1, n // 2] <= 1e-3


@pytest.mark.parametrize("case", R.MINING_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", R.MINING_MODES)
def test_mining(golden, case, mode):
    name, p, k, e, m, sigma, seed = case
    g = golden("mining")
    dist = g[f"{name}/D"]
    key = f"{name}/{mode}"
    np.random.seed(1000 + seed)                 # same legacy stream the reference consumed
    mining.sample_batch([k + 2] * p, p, k)      # the P*K sampling draws precede the mining draws
    out = mining.mine_triplets(dist, p, k, m, mode)
    np.testing.assert_array_equal(out["loss_values"], g[f"{key}/loss_values"])
    assert out["fallback"] == bool(g[f"{key}/fallback"])
    act = g[f"{key}/selected"] >= 0
    assert np.array_equal(out["selected"] >= 0, act)
    if mode == "hardest":
        assert np.array_equal(out["triplets"], g[f"{key}/triplets"])
        assert np.array_equal(out["selected"], g[f"{key}/selected"])
    else:
        # candidate sets are exact; the reference's picks lie inside them
        assert np.array_equal(out["candidates"], g[f"{key}/candidates"])
        sel = g[f"{key}/selected"]
        assert all(out["candidates"][i, s] for i, s in enumerate(sel) if s >= 0)
        # with the same RandomState stream the restatement reproduces the picks too
        assert np.array_equal(out["triplets"], g[f"{key}/triplets"])


@pytest.mark.parametrize("case", R.MINING_CASES[:3], ids=lambda c: c[0])
def test_mining_from_embeddings_matches_captured_matrix(golden, case):
    name, p, k, e, m, sigma, seed = case
    g = golden("mining")
    x = R.clustered_embeddings(seed, p, k, e, sigma)
    if g[f"{name}/X"].size:
        assert np.array_equal(g[f"{name}/X"], x)
    out = mining.mine_from_embeddings(x, p, k, m, "hardest")
    assert np.array_equal(out["triplets"], g[f"{name}/hardest/triplets"])


def test_batch_hard_properties():
    x = R.clustered_embeddings(3, 8, 4, 64, 0.3)
    d = pairwise.pairwise_distances(x)
    t = mining.batch_hard(d, 8, 4)
    assert t.shape == (32, 3) and np.array_equal(t[:, 0], np.arange(32))
    for a, p_, n_ in t:
        assert a // 4 == p_ // 4 and a != p_ and a // 4 != n_ // 4
        same = [j for j in range(32) if j // 4 == a // 4 and j != a]
        other = [j for j in range(32) if j // 4 != a // 4]
        assert d[a, p_] == max(d[a, same]) and d[a, n_] == min(d[a, other])


@pytest.mark.parametrize("case", R.KNN_CASES, ids=lambda c: c[0])
def test_knn_oracle_vs_sklearn_golden(golden, case):
    from oracle import knn as oknn
    name, nc, per, e, sigma, nq, seed = case
    g = golden("knn")
    x, y, q, qy = R.knn_data(nc, per, e, sigma, nq, seed)
    assert np.array_equal(qy, g[f"{name}/query_labels"])
    dist, idx = oknn.kneighbors(q, x, 5)
    assert np.array_equal(idx, g[f"{name}/k1/idx5"])
    np.testing.assert_allclose(dist, g[f"{name}/k1/dist5"], rtol=1e-6, atol=1e-6)
    for k in (1, 5):
        assert np.array_equal(oknn.predict(q, x, y, k), g[f"{name}/k{k}/predict"])
