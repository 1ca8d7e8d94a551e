#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference code.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  The reference's loss module and mining generator are
imported unmodified; TensorFlow / cv2 / albumentations are absent here, so
`tensorflow.keras.backend` is replaced in sys.modules by a NumPy-backed
stand-in that implements exactly the seven K.* calls the loss module makes
(square, maximum, sum, mean, equal, cast, epsilon) in float64.  Nothing from
the reference is copied into the fixtures: they hold inputs (or the recipe
seed for big inputs), and the reference's outputs.

Reference entry points exercised (file:line under /root/reference):
  embedding_net/losses_and_accuracies.py:4   contrastive_loss
  embedding_net/losses_and_accuracies.py:14  triplet_loss
  embedding_net/losses_and_accuracies.py:47  accuracy
  embedding_net/datagenerators.py:188-199    hardest/random_hard/semihard
  embedding_net/datagenerators.py:201-258    get_batch_triplets_mining
  embedding_net/datagenerators.py:219        sklearn pairwise_distances call

Usage:  python tests/golden/gen_golden.py      (writes next to this file)
"""
import os
import sys
import types
from unittest import mock

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipes as R  # noqa: E402

REF = "/root/reference"


# --------------------------------------------------------------------------
# NumPy stand-in for tensorflow.keras.backend (float64 arithmetic)
# --------------------------------------------------------------------------
class _Shape(tuple):
    def as_list(self):
        return list(self)


class KT:
    """Just enough tensor for losses_and_accuracies.py."""

    def __init__(self, a):
        self.a = np.asarray(a)

    @property
    def shape(self):
        return _Shape(self.a.shape)

    @property
    def dtype(self):
        return self.a.dtype

    def __getitem__(self, idx):
        return KT(self.a[idx])

    @staticmethod
    def _u(x):
        return x.a if isinstance(x, KT) else x

    def __add__(self, o): return KT(self.a + self._u(o))
    def __radd__(self, o): return KT(self._u(o) + self.a)
    def __sub__(self, o): return KT(self.a - self._u(o))
    def __rsub__(self, o): return KT(self._u(o) - self.a)
    def __mul__(self, o): return KT(self.a * self._u(o))
    def __rmul__(self, o): return KT(self._u(o) * self.a)
    def __lt__(self, o): return KT(self.a < self._u(o))


def _install_stubs():
    u = KT._u
    K = types.ModuleType("tensorflow.keras.backend")
    K.square = lambda x: KT(np.square(u(x)))
    K.maximum = lambda x, y: KT(np.maximum(u(x), u(y)))
    K.sum = lambda x, axis=None, keepdims=False: KT(np.sum(u(x), axis=axis, keepdims=keepdims))
    K.mean = lambda x, axis=None: KT(np.mean(u(x).astype(np.float64), axis=axis))
    K.equal = lambda x, y: KT(np.equal(u(x), u(y)))
    K.cast = lambda x, dtype: KT(u(x).astype(dtype))
    K.epsilon = lambda: 1e-7

    tf = types.ModuleType("tensorflow")
    keras = types.ModuleType("tensorflow.keras")
    kutils = types.ModuleType("tensorflow.keras.utils")
    kutils.Sequence = object
    keras.backend = K
    keras.utils = kutils
    keras.optimizers = mock.MagicMock()
    tf.keras = keras
    mods = {
        "tensorflow": tf,
        "tensorflow.keras": keras,
        "tensorflow.keras.backend": K,
        "tensorflow.keras.utils": kutils,
        "tensorflow.keras.optimizers": keras.optimizers,
        "cv2": mock.MagicMock(),
        "albumentations": mock.MagicMock(),
        "matplotlib": mock.MagicMock(),
        "matplotlib.pyplot": mock.MagicMock(),
        "sklearn.manifold": mock.MagicMock(),
    }
    sys.modules.update(mods)
    sys.path.insert(0, REF)


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


SMALL = 300 * 1024  # inputs up to this many bytes are stored; else recipe only


def gen_losses(lac):
    out = {}
    for name, t, e, m, kind, seed in R.TRIPLET_CASES:
        y = R.triplet_rows(seed, t, e, kind)
        loss = lac.triplet_loss(m)(None, KT(y.astype(np.float64))).a
        assert loss.shape == (t,)
        out[f"{name}/loss"] = loss
        out[f"{name}/margin"] = np.float64(m)
        if y.nbytes <= SMALL:
            out[f"{name}/y_pred"] = y
    y = R.triplet_edge_rows()
    out["edge/y_pred"] = y
    out["edge/margin"] = np.float64(0.5)
    out["edge/loss"] = lac.triplet_loss(0.5)(None, KT(y.astype(np.float64))).a
    _save("triplet_loss", **out)

    out = {}
    for name, b, hi, seed in R.SIAMESE_CASES:
        y, d = R.siamese_pairs(seed, b, hi)
        out[f"{name}/y_true"], out[f"{name}/y_pred"] = y, d
        out[f"{name}/contrastive"] = np.float64(
            lac.contrastive_loss(KT(y.astype(np.float64)), KT(d.astype(np.float64))).a)
        out[f"{name}/accuracy"] = np.float64(lac.accuracy(KT(y), KT(d)).a)
    # edges: d in {0, 0.5, 1} for both labels
    d = np.array([[0.0], [0.5], [1.0], [0.0], [0.5], [1.0]], np.float32)
    y = np.array([[1], [1], [1], [0], [0], [0]], np.float32)
    out["edge/y_true"], out["edge/y_pred"] = y, d
    out["edge/contrastive"] = np.float64(
        lac.contrastive_loss(KT(y.astype(np.float64)), KT(d.astype(np.float64))).a)
    out["edge/accuracy"] = np.float64(lac.accuracy(KT(y), KT(d)).a)
    _save("siamese_losses", **out)


def gen_pairwise(dg):
    # the exact callable the reference imported at datagenerators.py:7
    pairwise_distances = dg.pairwise_distances
    out = {}
    for name, n, e, seed, dup in R.PAIRWISE_CASES:
        x = R.pairwise_input(n, e, seed, dup)
        d = pairwise_distances(x)
        assert d.dtype == np.float32 and d.shape == (n, n)
        out[f"{name}/D"] = d
        if x.nbytes <= SMALL:
            out[f"{name}/X"] = x
    _save("pairwise_distances", **out)


class _FakeModel:
    """embedding_model.predict(images) -> rows of a fixed matrix, selected by
    the row id the patched _get_images_set wrote into the image."""

    def __init__(self, x):
        self.x = x

    def predict(self, images):
        rows = images[:, 0, 0, 0].astype(np.int64)
        return self.x[rows]


def gen_mining(dg):
    out = {}
    for name, p, k, e, m, sigma, seed in R.MINING_CASES:
        x = R.clustered_embeddings(seed, p, k, e, sigma)
        n = p * k
        for mode in R.MINING_MODES:
            classes = [f"c{i}" for i in range(p)]
            files = {c: list(range(k + 2)) for c in classes}
            gen = dg.TripletsDataGenerator(
                embedding_model=_FakeModel(x), class_files_paths=files,
                class_names=classes, input_shape=(1, 1, 3), k_classes=p,
                k_samples=k, margin=m, negatives_selection_mode=mode)

            counter = {"row": 0}

            def fake_images(clsss, idxs, with_aug=True, _c=counter, _k=k):
                rows = np.arange(_c["row"], _c["row"] + _k, dtype=np.float64)
                _c["row"] += _k
                return np.broadcast_to(rows[:, None, None, None], (_k, 1, 1, 3)).copy()

            gen._get_images_set = fake_images

            log_loss, log_sel, log_cand = [], [], []
            orig_fn = gen.negative_selection_fn

            def logged_fn(loss_values, margin=0.5, _f=orig_fn):
                log_loss.append(np.array(loss_values))
                r = _f(loss_values, margin=margin)
                log_sel.append(-1 if r is None else int(r))
                return r

            gen.negative_selection_fn = logged_fn

            real_choice = np.random.choice

            def logged_choice(a, *args, **kw):
                if not args and not kw and isinstance(a, np.ndarray):
                    log_cand.append((len(log_loss) - 1, a.copy()))
                return real_choice(a, *args, **kw)

            captured = {}
            real_pd = dg.pairwise_distances

            def logged_pd(emb):
                captured["emb"] = np.array(emb)
                captured["D"] = real_pd(emb)
                return captured["D"]

            np.random.seed(1000 + seed)
            with mock.patch.object(np.random, "choice", logged_choice), \
                    mock.patch.object(dg, "pairwise_distances", logged_pd):
                (ta, tp, tn), targets = gen.get_batch_triplets_mining()

            assert np.array_equal(captured["emb"], x), "fake model must see rows in order"
            trip = np.stack([ta[:, 0, 0, 0], tp[:, 0, 0, 0], tn[:, 0, 0, 0]], 1).astype(np.int32)
            npairs = p * k * (k - 1) // 2
            assert len(log_loss) == npairs
            cand = np.zeros((npairs, n - k), bool)
            for pair_idx, c in log_cand:
                cand[pair_idx, c] = True
            key = f"{name}/{mode}"
            out[f"{key}/triplets"] = trip
            out[f"{key}/targets"] = np.asarray(targets)
            out[f"{key}/loss_values"] = np.stack(log_loss).astype(np.float32)
            out[f"{key}/selected"] = np.asarray(log_sel, np.int32)
            out[f"{key}/candidates"] = cand
            out[f"{key}/fallback"] = np.bool_(all(s < 0 for s in log_sel))
            print(f"    {key}: T={len(trip)} active={sum(s >= 0 for s in log_sel)}/{npairs}"
                  f" fallback={bool(out[f'{key}/fallback'])}")
        out[f"{name}/X"] = x if x.nbytes <= SMALL else np.zeros((0,), np.float32)
        out[f"{name}/D"] = captured["D"]
        out[f"{name}/pkem"] = np.array([p, k, e, m, sigma, seed], np.float64)
    _save("mining", **out)


def gen_knn(models_src_check=True):
    """The reference only names sklearn's KNeighborsClassifier (models.py:15,134-137); golden = real sklearn."""
    from sklearn.neighbors import KNeighborsClassifier
    out = {}
    for name, nc, per, e, sigma, nq, seed in R.KNN_CASES:
        x, y, q, qy = R.knn_data(nc, per, e, sigma, nq, seed)
        for k in (1, 5):
            clf = KNeighborsClassifier(n_neighbors=k).fit(x, y)
            dist, idx = clf.kneighbors(q, n_neighbors=5)
            out[f"{name}/k{k}/predict"] = clf.predict(q)
            out[f"{name}/k{k}/dist5"], out[f"{name}/k{k}/idx5"] = dist.astype(np.float32), idx
        out[f"{name}/query_labels"] = qy
    _save("knn", **out)


def main():
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs /root/reference (build container only)")
    _install_stubs()
    from embedding_net import losses_and_accuracies as lac
    from embedding_net import datagenerators as dg
    print("reference imported from", os.path.dirname(lac.__file__))
    gen_losses(lac)
    gen_pairwise(dg)
    gen_mining(dg)
    gen_knn()


if __name__ == "__main__":
    main()
