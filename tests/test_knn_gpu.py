"""GPU parity of the post-training evaluation path: query-gallery distances, top-k,
kNN vote and the EmbeddingNet encodings / predict_knn / accuracy surface, vs real-sklearn golden vectors."""
import numpy as np
import pytest
import torch

import recipes as R
from oracle import knn as oknn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", R.KNN_CASES, ids=lambda c: c[0])
def test_knn_kernels_vs_sklearn_golden(golden, case):
    from embeddingnet_amd import ops
    from embeddingnet_amd.knn import KNNClassifier
    name, nc, per, e, sigma, nq, seed = case
    g = golden("knn")
    x, y, q, qy = R.knn_data(nc, per, e, sigma, nq, seed)
    d = ops.cross_distances(torch.tensor(q, device=DEV), torch.tensor(x, device=DEV)).cpu().numpy()
    np.testing.assert_allclose(d, oknn.cross_distances(q, x), rtol=1e-4, atol=2e-3)      # as the NxN matrix
    for k in (1, 5):
        clf = KNNClassifier(n_neighbors=k, device=torch.device(DEV)).fit(x, list(y))
        dist, idx = clf.kneighbors(q, n_neighbors=5)
        ref_idx, ref_dist = g[f"{name}/k{k}/idx5"], g[f"{name}/k{k}/dist5"]
        np.testing.assert_allclose(dist, ref_dist, rtol=1e-4, atol=2e-3)
        # neighbour identity: equal unless two gallery rows are within fp32 noise of each other for that query
        for r in range(nq):
            for c in range(5):
                if idx[r, c] != ref_idx[r, c]:
                    assert abs(ref_dist[r, c] - oknn.cross_distances(q[r:r + 1], x[idx[r, c]:idx[r, c] + 1])[0, 0]) < 1e-5
        pred = clf.predict(q)
        agree = np.mean(pred == g[f"{name}/k{k}/predict"])
        assert agree >= 0.99, agree
    # selection and vote are exact integer work on a given matrix
    dm = torch.tensor(oknn.cross_distances(q, x), device=DEV)
    val, idx = ops.topk_smallest(dm, 5)
    ref_val, ref_idx = oknn.kneighbors(q, x, 5)
    assert np.array_equal(idx.cpu().numpy(), ref_idx) and np.array_equal(val.cpu().numpy(), ref_val)
    vote = ops.knn_vote(idx, torch.tensor(y, dtype=torch.int32, device=DEV)).cpu().numpy()
    assert np.array_equal(vote, np.array([np.bincount(y[r]).argmax() for r in ref_idx]))


def test_encodings_and_knn_accuracy_end_to_end(tmp_path):
    """Train simple2 briefly on a synthetic 6-class set, export encodings, fit the kNN and score the held-out images."""
    from embeddingnet_amd.datagenerators import SyntheticDataLoader
    from embeddingnet_amd.models import TripletNet
    from embeddingnet_amd.train_step import TripletTrainer
    dev = torch.device(DEV)
    params = {"model": dict(input_shape=[64, 64, 3], encodings_len=64, mode="triplet", distance_type="l2",
                            backbone_name="simple2", backbone_weights=None, freeze_backbone=False,
                            embeddings_normalization=True, device=dev, seed=0),
              "dataloader": {}, "generator": {}, "train": {}, "general": {"work_dir": str(tmp_path), "project_name": "p"}}
    data = SyntheticDataLoader(6, 16, (64, 64, 3), noise=0.2, validate=True, val_ratio=0.25, seed=3)
    net = TripletNet(params, training=True)
    opt = torch.optim.Adam(net.base_model.parameters(), lr=1e-3, eps=1e-7)
    tr = TripletTrainer(net.base_model, opt, 6, 4, margin=0.5, negatives_selection_mode="semihard", seed=1)
    rs = np.random.RandomState(0)
    for _ in range(30):
        batch = np.concatenate([data.train_data[c][rs.choice(12, 4, replace=False)] for c in data.class_names])
        tr.step(torch.from_numpy(batch).to(dev))
    enc = net.generate_encodings(data, max_n_samples=10, shuffle=False)
    assert enc["encodings"].shape == (60, 64) and len(enc["labels"]) == 60 and len(enc["paths"]) == 60
    net.save_encodings(enc, save_folder=str(tmp_path))
    clf = net.load_encodings(str(tmp_path / "encodings.pkl"), knn_k=1)
    acc = net.calculate_prediction_accuracy(data)
    assert acc["top1"] >= 0.9 and acc["top5"] >= acc["top1"], acc
    img = data.val_data[data.class_names[2]][0]
    label, top5 = net.predict_knn(img, with_top5=True)
    assert label[0] == data.class_names[2] and len(top5) == 5
    # the classifier agrees with the oracle on the same encodings
    q = net.base_model.predict(data.val_data[data.class_names[0]])
    lookup = {c: i for i, c in enumerate(sorted(set(enc["labels"])))}
    yi = np.array([lookup[l] for l in enc["labels"]])
    ref = oknn.predict(q, enc["encodings"], yi, 1)
    assert np.array_equal(np.array([lookup[l] for l in clf.predict(q)]), ref)


def test_softmax_cross_entropy_and_pretraining(tmp_path):
    """the softmax head's loss/metric vs the oracle, then a short pre-training run learns."""
    from embeddingnet_amd import backbones as B, ops
    from embeddingnet_amd.datagenerators import SimpleDataGenerator, SyntheticDataLoader
    from embeddingnet_amd.utils import get_optimizer
    from oracle import losses as olosses
    rs = np.random.RandomState(0)
    for b, c in [(8, 10), (33, 107), (5, 3)]:
        z = (rs.randn(b, c) * 3).astype(np.float32)
        t = np.eye(c, dtype=np.float32)[rs.randint(0, c, b)]
        zt = torch.tensor(z, device=DEV, requires_grad=True)
        loss, acc, prob = ops.softmax_cross_entropy(zt, torch.tensor(t, device=DEV))
        rl, ra, rp, rg = olosses.softmax_cross_entropy(z, t)
        np.testing.assert_allclose(loss.item(), rl, rtol=2e-6)
        assert acc.item() == pytest.approx(ra)
        np.testing.assert_allclose(prob.cpu().numpy(), rp, rtol=2e-5, atol=1e-7)
        (loss * 2).backward()
        np.testing.assert_allclose(zt.grad.cpu().numpy(), 2 * rg, rtol=2e-5, atol=1e-8)
    data = SyntheticDataLoader(5, 16, (64, 64, 3), noise=0.15, validate=True, val_ratio=0.25, seed=1)
    gen = SimpleDataGenerator(data.train_data, data.class_names, input_shape=[64, 64, 3], batch_size=6, n_batches=3)
    (x,), t = gen[0]
    assert x.shape == (6, 64, 64, 3) and t.shape == (6, 5) and np.all(t.sum(1) == 1) and len(gen) == 3
    base, backbone = B.get_backbone((64, 64, 3), encodings_len=32, backbone_name="resnet18", backbone_weights=None,
                                    device=torch.device(DEV))
    params_softmax = dict(optimizer=get_optimizer("adam", 1e-3), learning_rate=1e-3, decay_factor=0.99, step_size=1,
                          input_shape=[64, 64, 3], batch_size=20, val_steps=2, steps_per_epoch=8, n_epochs=4,
                          augmentations=None)
    hist = B.pretrain_backbone_softmax(backbone, data, params_softmax, {"work_dir": str(tmp_path), "project_name": "pre"})
    assert hist["loss"][-1] < hist["loss"][0] and hist["accuracy"][-1] > 0.5
    import os
    assert os.listdir(tmp_path / "pre" / "pretraining_model" / "weights")
