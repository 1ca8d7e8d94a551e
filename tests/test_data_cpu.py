"""CPU tests of the input side of the hot path: ENDataLoader (directory tree, CSV, Google-Landmarks
layout), get_image (decode, resize, BGR), and the batch layouts the generators hand to the training step
(reference datagenerators.py:16-111, 145-156, 264-418; utils.py:13-21).  No GPU: these are host-side."""
import os

import numpy as np
import pytest


def _png(path, rgb, size=(20, 12)):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.new("RGB", size, rgb).save(path)


@pytest.fixture()
def tree(tmp_path):
    """3 classes x 10 images; class 'c2' keeps its images one level deeper; plus files the reference skips."""
    root = tmp_path / "data"
    for ci, cl in enumerate(["c0", "c1", "c2"]):
        for i in range(10):
            sub = "part%d/" % (i % 2) if cl == "c2" else ""
            _png(str(root / cl / (sub + "img%02d.%s" % (i, "png" if i % 2 else "jpg"))), (10 * i, 40 + 20 * ci, 200))
    (root / "c0" / "notes.txt").write_text("not an image")
    _png(str(root / "c0" / "._hidden.png"), (0, 0, 0))
    (root / "readme.md").write_text("not a class")
    return root


def test_get_image_decodes_resizes_and_returns_bgr(tmp_path):
    from embeddingnet_amd.datagenerators import get_image
    _png(str(tmp_path / "a.png"), (10, 20, 30), size=(20, 12))
    img = get_image(str(tmp_path / "a.png"))
    assert img.dtype == np.uint8 and img.shape == (12, 20, 3)
    assert tuple(img[0, 0]) == (30, 20, 10)                               # cv2.imread order: B, G, R
    small = get_image(str(tmp_path / "a.png"), [8, 6, 3])
    assert small.shape == (6, 8, 3)                                       # cv2.resize(img, (input_shape[0], input_shape[1])) = (w, h)
    assert get_image(str(tmp_path / "missing.png")) is None


def test_loader_from_directory(tree):
    from embeddingnet_amd.datagenerators import ENDataLoader
    dl = ENDataLoader(str(tree), validate=True, val_ratio=0.2)
    assert dl.class_names == ["c0", "c1", "c2"] and dl.n_classes == 3
    assert dl.n_samples == {"c0": 10, "c1": 10, "c2": 10}                  # txt and '._' files skipped, sub-dirs walked
    assert all(len(dl.train_data[c]) == 8 and len(dl.val_data[c]) == 2 for c in dl.class_names)
    assert not set(dl.train_data["c1"]) & set(dl.val_data["c1"])
    again = ENDataLoader(str(tree), validate=True, val_ratio=0.2)          # train_test_split(random_state=42)
    assert again.val_data == dl.val_data
    flat = ENDataLoader(str(tree), validate=False)
    assert flat.val_data == {} and flat.train_data == flat.class_files_paths


def test_loader_from_csv_and_google_layout(tree, tmp_path):
    import pandas as pd
    from embeddingnet_amd.datagenerators import ENDataLoader
    rows = [("c1/img%02d.%s" % (i, "png" if i % 2 else "jpg"), "beta") for i in range(6)] + \
           [("c0/img%02d.%s" % (i, "png" if i % 2 else "jpg"), "alpha") for i in range(4)]
    csv = tmp_path / "train.csv"
    pd.DataFrame(rows, columns=["image_id", "label"]).to_csv(csv, index=False)
    dl = ENDataLoader(str(tree), train_csv_file=str(csv), validate=False)
    assert dl.class_names == ["beta", "alpha"]                              # order of first appearance (:74)
    assert dl.n_samples == {"beta": 6, "alpha": 4}
    assert all(os.path.exists(p) for p in dl.train_data["alpha"])
    vcsv = tmp_path / "val.csv"
    pd.DataFrame(rows[:2] + rows[-2:], columns=["image_id", "label"]).to_csv(vcsv, index=False)
    dv = ENDataLoader(str(tree), train_csv_file=str(csv), val_csv_file=str(vcsv), validate=True)
    assert dv.train_data == dv.class_files_paths and {k: len(v) for k, v in dv.val_data.items()} == {"beta": 2, "alpha": 2}
    gcsv = tmp_path / "g.csv"
    pd.DataFrame([("abcdef", 7), ("a1c9", 7)], columns=["id", "landmark_id"]).to_csv(gcsv, index=False)
    dg = ENDataLoader("/data/gl", train_csv_file=str(gcsv), image_id_column="id", label_column="landmark_id",
                      validate=False, is_google=True)
    assert dg.train_data == {"7": ["/data/gl/a/b/c/abcdef.jpg", "/data/gl/a/1/c/a1c9.jpg"]}


def test_images_set_contract(tree):
    """_get_images_set: float32 [n,H,W,3] in [0,1], BGR, resized to input_shape (reference :145-156)."""
    from embeddingnet_amd.datagenerators import ENDataGenerator, ENDataLoader
    dl = ENDataLoader(str(tree), validate=False)
    gen = ENDataGenerator(dl.train_data, dl.class_names, input_shape=[16, 16, 3])
    x = gen._get_images_set("c1", [1, 3])
    assert x.dtype == np.float32 and x.shape == (2, 16, 16, 3) and 0 <= x.min() and x.max() <= 1
    np.testing.assert_allclose(x[0, 0, 0], np.array([200, 60, 10]) / 255., atol=1e-6)      # B, G, R of img01.png
    mixed = gen._get_images_set(["c0", "c2"], [1, 1])
    np.testing.assert_allclose(mixed[:, 0, 0, 1], np.array([40, 80]) / 255., atol=1e-6)


def test_generator_batch_layouts(tree):
    from embeddingnet_amd.datagenerators import (ENDataLoader, SiameseDataGenerator, SimpleDataGenerator,
                                                 SimpleTripletsDataGenerator)
    dl = ENDataLoader(str(tree), validate=False)
    kw = dict(input_shape=[16, 16, 3], batch_size=8, n_batches=3)
    (x1, x2), y = SiameseDataGenerator(dl.train_data, dl.class_names, **kw)[0]
    assert x1.shape == x2.shape == (8, 16, 16, 3) and y.tolist() == [1, 1, 1, 1, 0, 0, 0, 0]   # :355-374
    cls_of = lambda v: np.round((v[:, 0, 0, 1] * 255 - 40) / 20).astype(int)   # green encodes the class (40 + 20 ci; jpg is lossy)
    assert np.array_equal(cls_of(x2[:4]), cls_of(x1[:4])) and not np.any(cls_of(x2[4:]) == cls_of(x1[4:]))
    val = SiameseDataGenerator(dl.train_data, dl.class_names, val_gen=True, n_batches_val=5, **kw)
    assert len(val) == 5
    (a, p, n), t = SimpleTripletsDataGenerator(dl.train_data, dl.class_names, **kw)[0]
    assert a.shape == p.shape == n.shape == (8, 16, 16, 3) and t.shape == (8,) and np.all(t == 1)
    assert np.array_equal(cls_of(a), cls_of(p)) and not np.any(cls_of(a) == cls_of(n))
    (x,), onehot = SimpleDataGenerator(dl.train_data, dl.class_names, **kw)[0]
    assert x.shape == (8, 16, 16, 3) and onehot.shape == (8, 3) and np.all(onehot.sum(1) == 1)
    assert np.array_equal(onehot.argmax(1), cls_of(x))


def test_triplet_sampler_is_class_contiguous_with_replacement(tree):
    """sample_batch(): P classes without replacement, K images per class WITH replacement, class blocks contiguous
    (reference :202-205, 226-227) — host-side part of TripletsDataGenerator; no model needed."""
    from embeddingnet_amd.datagenerators import ENDataLoader, TripletsDataGenerator
    dl = ENDataLoader(str(tree), validate=False)
    gen = TripletsDataGenerator(embedding_model=None, class_files_paths=dl.train_data, class_names=dl.class_names,
                                input_shape=[16, 16, 3], k_classes=3, k_samples=12, margin=0.5,
                                negatives_selection_mode="semihard")
    np.random.seed(1)
    batch = gen.sample_batch()
    assert batch.shape == (36, 16, 16, 3)
    cls = np.round((batch[:, 0, 0, 1] * 255 - 40) / 20).astype(int).reshape(3, 12)
    assert (cls == cls[:, :1]).all() and len(set(cls[:, 0])) == 3          # contiguous blocks, distinct classes
    with pytest.raises(KeyError):
        TripletsDataGenerator(None, dl.train_data, dl.class_names, negatives_selection_mode="nope")


def _gen(tree, k_classes=3, k_samples=4, shape=(16, 16, 3)):
    from embeddingnet_amd.datagenerators import ENDataLoader, TripletsDataGenerator
    dl = ENDataLoader(str(tree), validate=False)
    return TripletsDataGenerator(embedding_model=None, class_files_paths=dl.train_data, class_names=dl.class_names,
                                 input_shape=list(shape), k_classes=k_classes, k_samples=k_samples, margin=0.5,
                                 negatives_selection_mode="semihard")


def test_sample_plan_draws_the_reference_stream_and_loads_the_same_batch(tree):
    """sample_batch() = load_plan(sample_plan()): the plan consumes the global np.random stream exactly as the reference's
    sampler does (datagenerators.py:202-205: one choice() of the classes, then one choice() per class), and the uint8 form
    the input pipeline moves (load_plan_u8) divided by 255 in float32 IS the float batch."""
    gen = _gen(tree)
    np.random.seed(5)
    want_cls = np.random.choice(gen.n_classes, size=3, replace=False)
    want_idx = [np.random.choice(gen.n_samples[gen.class_names[c]], size=4, replace=True) for c in want_cls]
    after = np.random.randint(1 << 30)
    np.random.seed(5)
    classes, idxs = gen.sample_plan()
    assert classes == [gen.class_names[c] for c in want_cls] and all(np.array_equal(a, b) for a, b in zip(idxs, want_idx))
    assert np.random.randint(1 << 30) == after                             # nothing else was drawn
    np.random.seed(5)
    batch = gen.sample_batch()
    u8 = gen.load_plan_u8((classes, idxs))
    assert u8.dtype == np.uint8 and u8.shape == (12, 16, 16, 3)
    assert np.array_equal(u8.astype(np.float32) / np.float32(255.), batch)


def test_prefetcher_delivers_the_planned_batches_in_order(tree):
    """BatchPrefetcher's host side: plans are drawn on the consumer's thread (so the np.random stream is the sequential one),
    worker threads decode `depth` batches ahead, batches come out in plan order and equal the sequential sample_batch() run."""
    from embeddingnet_amd.input_pipeline import BatchPrefetcher
    gen = _gen(tree)
    np.random.seed(11)
    want = [gen.sample_batch() for _ in range(9)]
    np.random.seed(11)
    pf = BatchPrefetcher(gen.sample_plan, gen.load_plan_u8, (12, 16, 16, 3), "cpu", depth=4, workers=3)
    try:
        for w in want[: 9 - 4]:                                            # (the prefetcher has drawn `depth` plans ahead)
            got = pf.next_u8().numpy().astype(np.float32) / np.float32(255.)
            assert np.array_equal(got, w)
        with pytest.raises(Exception):
            pf.next()                                                      # device tensors need the device: no CPU conversion
    finally:
        pf.close()


def test_prefetcher_surfaces_a_worker_error(tree):
    from embeddingnet_amd.input_pipeline import BatchPrefetcher
    gen = _gen(tree)
    calls = []

    def load(plan, out):
        calls.append(1)
        if len(calls) == 3:
            raise FileNotFoundError("gone.jpg")
        return gen.load_plan_u8(plan, out)
    pf = BatchPrefetcher(gen.sample_plan, load, (12, 16, 16, 3), "cpu", depth=2, workers=1)
    try:
        pf.next_u8(); pf.next_u8()
        with pytest.raises(FileNotFoundError):
            pf.next_u8()
    finally:
        pf.close()


def test_prefetcher_with_decode_worker_processes(tree):
    """The same batches when worker PROCESSES (python -m embeddingnet_amd._decode_worker: no torch) decode into the shared
    staging array; a missing file surfaces as an error on the consumer's thread; the staging file is removed on close."""
    from embeddingnet_amd.input_pipeline import BatchPrefetcher
    gen = _gen(tree)
    np.random.seed(13)
    want = [gen.sample_batch() for _ in range(7)]
    np.random.seed(13)
    pf = BatchPrefetcher(gen.sample_plan, gen.load_plan_u8, (12, 16, 16, 3), "cpu", depth=3, workers=2, paths_fn=gen.plan_paths,
                         input_shape=gen.input_shape, rows_per_task=5)
    path = pf.procs.path
    try:
        assert os.path.exists(path)
        for w in want[:4]:
            assert np.array_equal(pf.next_u8().numpy().astype(np.float32) / np.float32(255.), w)
    finally:
        pf.close()
    assert not os.path.exists(path)
    bad = BatchPrefetcher(gen.sample_plan, gen.load_plan_u8, (12, 16, 16, 3), "cpu", depth=2, workers=2,
                          paths_fn=lambda plan: ["/nonexistent/x.jpg"] * 12, input_shape=gen.input_shape)
    try:
        with pytest.raises(RuntimeError, match="FileNotFoundError"):
            bad.next_u8()
    finally:
        bad.close()
