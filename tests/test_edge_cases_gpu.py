"""GPU edge cases and size-independent properties (through the C ABI): minimal and ragged shapes,
config-maximum sizes, bitwise repeatability, error paths."""
import numpy as np
import pytest
import torch

import recipes as R
from oracle import losses as olosses
from oracle import mining as omining
from oracle import pairwise as opair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def test_minimal_shapes():
    from embeddingnet_amd import layers as L
    from embeddingnet_amd import ops
    from embeddingnet_amd.losses_and_accuracies import accuracy, contrastive_loss, triplet_loss
    assert ops.pairwise_distances(g(np.ones((1, 5)))).cpu().numpy().tolist() == [[0.0]]
    y = np.array([[1, 0, 0, 1, 1, 0]], np.float32)              # T=1, E=2: pos 2, neg 0 -> 2.5
    assert triplet_loss(0.5)(None, g(y)).item() == 2.5
    assert contrastive_loss(g([[1.0]]), g([[0.5]])).item() == 0.25
    assert accuracy(g([[0.0]]), g([[0.7]])).item() == 1.0
    # smallest mining problem: 2 classes x 2 samples (one pair per class, two negatives each)
    x = R.clustered_embeddings(0, 2, 2, 16, 0.3)
    d = opair.pairwise_distances(x)
    for mode in R.MINING_MODES:
        trip, count, sel = ops.mine_triplets(g(d), 2, 2, 0.5, mode, seed=3)
        ref = omining.mine_triplets(d, 2, 2, 0.5, mode, rng=np.random.RandomState(0))
        assert int(count.item()) == len(ref["triplets"])
        assert np.array_equal(sel.cpu().numpy() >= 0, ref["selected"] >= 0)
    # 1x1 image through a 1x1 conv, a batch-norm over a single row, a pool that covers the whole map
    conv = L.Conv2D(3, 5, 1).to(DEV)
    out = conv(g(np.ones((1, 1, 1, 3))))
    np.testing.assert_allclose(out.cpu().detach().numpy().ravel(),
                               conv.kernel.detach().cpu().numpy().sum(2).ravel(), rtol=1e-6)
    bn = L.BatchNormalization(4).to(DEV).train()
    # zero variance: y = x*rstd - mean*rstd with rstd = 1/sqrt(eps) ~ 31.6, so |y| <= a few ulp of 95
    assert torch.allclose(bn(g(np.full((1, 1, 1, 4), 3.0))), torch.zeros(1, 1, 1, 4, device=DEV), atol=2e-5)
    assert L.MaxPool2D(2)(g(np.arange(4.0).reshape(1, 2, 2, 1))).item() == 3.0


def test_error_paths_raise():
    from embeddingnet_amd import _lib, layers as L, ops
    with pytest.raises(_lib.EmbnetError):
        ops.mine_triplets(g(np.zeros((6, 6))), 3, 3, 0.5, "hardest")          # matrix is not (p*k)^2
    with pytest.raises(_lib.EmbnetError, match="k_classes>=2"):
        ops.mine_triplets(g(np.zeros((3, 3))), 1, 3, 0.5, "hardest")
    with pytest.raises(_lib.EmbnetError, match="not 3\\*E"):
        ops.triplet_hinge(g(np.zeros((2, 7))), 0.5)
    with pytest.raises(_lib.EmbnetError, match="does not fit"):
        L.Conv2D(3, 8, 10).to(DEV)(g(np.zeros((1, 5, 5, 3))))                  # 'simple' below its minimum size
    with pytest.raises(_lib.EmbnetError, match="channels"):
        L.Conv2D(4, 8, 3).to(DEV)(g(np.zeros((1, 8, 8, 3))))
    with pytest.raises(KeyError):
        ops.mine_triplets(g(np.zeros((4, 4))), 2, 2, 0.5, "batch_all")


def test_mining_at_maximum_config_size_properties():
    """C5-sized and larger (N = 256, 1024, 4096; K = 4): every triplet is class-valid, pair order is the
    reference's, the hardest negative really is the closest other-class row, counts match the oracle."""
    from embeddingnet_amd import ops
    for p, e, sigma in [(64, 512, 0.25), (256, 256, 0.3), (1024, 128, 0.35)]:
        k, n, m = 4, p * 4, 0.5
        x = R.clustered_embeddings(5, p, k, e, sigma)
        d = ops.pairwise_distances(g(x))
        trip, count, sel = ops.mine_triplets(d, p, k, m, "hardest")
        t = int(count.item())
        tr = trip.cpu().numpy()[:t]
        dn = d.cpu().numpy()
        a, pp, ng = tr[:, 0], tr[:, 1], tr[:, 2]
        assert np.all(a // k == pp // k) and np.all(a < pp) and np.all(a // k != ng // k)
        key = a.astype(np.int64) * n + pp
        assert np.all(np.diff(key) > 0)                                     # combinations() order, no duplicates
        other = dn[a].copy()
        other[np.arange(t)[:, None], (a // k * k)[:, None] + np.arange(k)[None, :]] = np.inf
        assert np.array_equal(other.argmin(1), ng)                          # first arg-min == first arg-max of loss
        assert np.all(dn[a, pp] - dn[a, ng] + np.float32(m) > 0)
        ref = omining.mine_triplets(dn, p, k, m, "hardest")
        assert t == len(ref["triplets"]) and np.array_equal(tr, ref["triplets"])
        # batch-hard: one triplet per anchor, farthest positive / closest negative
        bh, cnt = ops.batch_hard(d, p, k)
        assert int(cnt.item()) == n and np.array_equal(bh.cpu().numpy(), omining.batch_hard(dn, p, k))


def test_step_is_bitwise_repeatable():
    """No float atomics anywhere: the same seed and batch give bit-identical losses and weights."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.train_step import TripletTrainer
    x = torch.rand((32, 64, 64, 3), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    runs = []
    for _ in range(2):
        base, _ = B.get_backbone((64, 64, 3), encodings_len=64, backbone_name="resnet18", backbone_weights=None,
                                 seed=5, device=torch.device(DEV))
        opt = torch.optim.SGD(base.parameters(), lr=0.05)
        tr = TripletTrainer(base, opt, 8, 4, margin=0.5, negatives_selection_mode="semihard", seed=11)
        losses = [tr.step(x).item() for _ in range(4)]
        runs.append((losses, torch.cat([p.detach().reshape(-1) for p in base.parameters()]).clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])


def test_loss_decreases_on_fixed_batch_all_modes():
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.train_step import TripletTrainer
    rs = np.random.RandomState(0)
    cls = rs.rand(8, 64, 64, 3)
    x = g(np.clip(np.repeat(cls, 4, axis=0) + 0.2 * rs.randn(32, 64, 64, 3), 0, 1))
    for mode in ("hardest", "semihard", "random_hard", "batch_hard"):
        base, _ = B.get_backbone((64, 64, 3), encodings_len=64, backbone_name="simple2", backbone_weights=None,
                                 seed=1, device=torch.device(DEV))
        opt = torch.optim.Adam(base.parameters(), lr=1e-3, eps=1e-7)
        tr = TripletTrainer(base, opt, 8, 4, margin=0.5, negatives_selection_mode=mode, seed=2)
        hist = [tr.step(x).item() for _ in range(25)]
        assert np.isfinite(hist).all() and min(hist[-5:]) < hist[0], (mode, hist[0], hist[-5:])


def test_large_hinge_backward_matches_dense_reference():
    """N=2048, T=3072: the row-owner gather backward equals a scatter-add reference."""
    from embeddingnet_amd import ops
    p, k, e = 512, 4, 64
    x = R.clustered_embeddings(2, p, k, e, 0.3)
    d = opair.pairwise_distances(x)
    emb = g(x).requires_grad_(True)
    trip, count, _ = ops.mine_triplets(g(d), p, k, 0.5, "hardest")
    mean, rows = ops.triplet_gather_loss(emb, trip, count, 0.5)
    mean.backward()
    t = int(count.item())
    tr = trip.cpu().numpy()[:t]
    y = np.concatenate([x[tr[:, 0]], x[tr[:, 1]], x[tr[:, 2]]], 1)
    gy = olosses.triplet_loss_grad(0.5, y, np.full(t, 1.0 / t))
    ref = np.zeros((p * k, e))
    np.add.at(ref, tr[:, 0], gy[:, :e])
    np.add.at(ref, tr[:, 1], gy[:, e:2 * e])
    np.add.at(ref, tr[:, 2], gy[:, 2 * e:])
    np.testing.assert_allclose(emb.grad.cpu().numpy(), ref, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(mean.item(), olosses.triplet_loss(0.5)(None, y).mean(), rtol=2e-5)
