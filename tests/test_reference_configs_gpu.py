"""The reference's two SHIPPED configurations on the HIP path: EfficientNet-B1 at 48x48 (20 x 3, semihard,
margin 0.5 — reference configs/road_signs_apollo.yml) and EfficientNet-B5 at 128x128 (3 x 3, semihard, margin 0.3 — reference
configs/template.yml), built from `parse_params` on committed YAMLs that carry the reference's keys and values
(configs/road_signs_efnb1.yml, configs/bengali_efnb5.yml).

  * the network the YAML names is built with the depth / width the efficientnet package derives for it (round_repeats,
    round_filters: B1 23 blocks, B5 39 blocks / 48-wide stem / 2048-wide top), forward embeddings and every parameter's
    gradient against the float64 oracle (oracle/backbones.py) at the configuration's own batch: 48x48 ends in 2x2 maps under
    5x5 'same' depthwise convs, 128x128 x 9 images in 4x4 maps with 144-sample BatchNorm statistics;
  * one fused training step (TripletTrainer: one forward, mining on the device, hinge, backward) with the YAML's mining
    rule and margin against the oracle composition: mined negatives from the reference's candidate sets, loss within 1e-4
    relative, gradients within the whole-backbone fp32 bound.
"""
import os

import numpy as np
import pytest
import torch

from oracle import backbones as OB

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {"b1": "configs/road_signs_efnb1.yml", "b5": "configs/bengali_efnb5.yml"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def g(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def _params(which):
    from embeddingnet_amd.utils import parse_params
    return parse_params(os.path.join(ROOT, CONFIGS[which]))


def _build(cfg, dev, seed):
    import warnings
    from embeddingnet_amd import backbones as B
    m = cfg["model"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")              # 'noisy-student' weights are not bundled: random initialisation
        base, backbone = B.get_backbone(tuple(m["input_shape"]), encodings_len=m["encodings_len"], backbone_name=m["backbone_name"],
                                        backbone_weights=m["backbone_weights"], freeze_backbone=m["freeze_backbone"],
                                        embeddings_normalization=m["embeddings_normalization"], seed=seed, device=dev)
    for mod in base.modules():
        if hasattr(mod, "enabled"):
            mod.enabled = False                      # dropout / drop-connect off for parity (random masks)
    return base, backbone


def _oracle_from(model, training, dtype=torch.float64):
    from embeddingnet_amd.backbones import keras_weights
    params = {k: v.detach().cpu().to(dtype).requires_grad_(v.requires_grad) for k, v in keras_weights(model).items()}
    return OB.Ctx(params, training=training)


@pytest.mark.parametrize("which,blocks,stem,top", [("b1", 23, 32, 1280), ("b5", 39, 48, 2048)])
def test_shipped_config_backbone_vs_oracle(dev, which, blocks, stem, top):
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import efficientnet as E
    cfg = _params(which)
    m, gen = cfg["model"], cfg["generator"]
    name, shape, enc = m["backbone_name"], tuple(m["input_shape"]), m["encodings_len"]
    bl, st, tp = E.block_list(name)
    assert (len(bl), st, tp) == (blocks, stem, top)
    batch = gen["k_classes"] * gen["k_samples"]                      # the reference's batch for this config: P x K
    base, backbone = _build(cfg, dev, seed=1)
    assert len([mod for mod in backbone.modules() if isinstance(mod, E.MBConv)]) == blocks
    rs = np.random.RandomState(0)
    x = rs.rand(batch, *shape).astype(np.float32)
    wgt = rs.randn(batch, enc).astype(np.float32)
    base.train()
    emb = base(g(x, dev))
    assert tuple(emb.shape) == (batch, enc)
    (emb * g(wgt, dev)).sum().backward()
    ctx = _oracle_from(base, True)
    embr = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc)
    ctx32 = _oracle_from(base, True, torch.float32)
    emb32 = OB.base_model(ctx32, torch.tensor(x), backbone_name=name, encodings_len=enc)
    # embeddings: within 5 x what the float32 oracle itself deviates from the float64 one (deep net, batch statistics on
    # few samples), and never more than 2e-3 of the largest component
    ref = embr.detach().numpy()
    floor = np.abs(emb32.detach().double().numpy() - ref).max() / np.abs(ref).max()
    err = np.abs(emb.detach().cpu().double().numpy() - ref).max() / np.abs(ref).max()
    assert err <= max(5 * floor, 5e-4) and err < 2e-3, (err, floor)
    (embr * torch.tensor(wgt, dtype=torch.float64)).sum().backward()
    (emb32 * torch.tensor(wgt)).sum().backward()
    got = B.keras_weights(base)
    bad, num, den, total = [], 0.0, 0.0, 0
    for k, p in ctx.params.items():
        if p.grad is None:
            continue
        total += 1
        assert got[k].grad is not None and torch.isfinite(got[k].grad).all(), k
        diff = got[k].grad.detach().cpu().double() - p.grad
        scale = max(p.grad.abs().max().item(), 1e-12)
        e = diff.abs().max().item() / scale
        fl = (ctx32.params[k].grad.double() - p.grad).abs().max().item() / scale
        num += (diff ** 2).sum().item()
        den += (p.grad ** 2).sum().item()
        if e >= 5 * fl + 1e-4:
            bad.append(f"{k}: {e:.2e} (floor {fl:.2e})")
    assert total > 4 * blocks
    assert len(bad) <= total // 40, bad
    assert (num / den) ** 0.5 < 2e-2
    # inference mode (moving statistics): what the reference's mining `predict` and the encodings export run
    pred = base.predict(x)
    ctx_i = _oracle_from(base, False)
    want = OB.base_model(ctx_i, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc).detach().numpy()
    assert np.abs(pred - want).max() / np.abs(want).max() < 5e-4


def _semihard_sets_ok(trip, emb, p, k, margin, tol=2e-5):
    """Every mined (a, p, n) is a pair the reference visits (datagenerators.py:231-234) with a negative from the pair's
    semihard candidate set (:196-199) on the oracle's distances; a pair with candidates (clear of the boundaries) has a triplet."""
    from oracle.pairwise import pairwise_distances
    d = pairwise_distances(emb)
    n = p * k
    gpu = {(int(a), int(b)): int(c) for a, b, c in trip}
    for c in range(p):
        lo, hi = c * k, (c + 1) * k
        neg = np.concatenate([np.arange(0, lo), np.arange(hi, n)])
        for i in range(lo, hi):
            for j in range(i + 1, hi):
                lv = d[i, j] - d[i, neg] + margin
                strict = (lv > tol) & (lv < margin - tol)
                loose = (lv > -tol) & (lv < margin + tol)
                got = gpu.get((i, j))
                if got is None:
                    assert not strict.any(), (i, j, "no triplet although the pair has semihard candidates")
                else:
                    assert loose[int(np.where(neg == got)[0][0])], (i, j, got)


@pytest.mark.parametrize("which", ["b1", "b5"])
def test_shipped_config_fused_step_vs_oracle(dev, which):
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.train_step import TripletTrainer
    cfg = _params(which)
    m, gen = cfg["model"], cfg["generator"]
    name, shape, enc = m["backbone_name"], tuple(m["input_shape"]), m["encodings_len"]
    p, k, margin, mode = gen["k_classes"], gen["k_samples"], gen["margin"], gen["negatives_selection_mode"]
    assert mode == "semihard"
    base, _ = _build(cfg, dev, seed=2)
    rs = np.random.RandomState(1)
    cls = rs.rand(p, *shape)
    x = np.clip(np.repeat(cls, k, axis=0) + 0.25 * rs.randn(p * k, *shape), 0, 1).astype(np.float32)
    tr = TripletTrainer(base, None, p, k, margin=margin, negatives_selection_mode=mode, seed=3)
    base.train()
    total, mean, count = tr.loss(g(x, dev))
    total.backward()
    trip = tr.last_triplets[0][: int(count.item())].cpu().numpy()
    W = {kk: v.detach().cpu().double().requires_grad_(v.requires_grad) for kk, v in B.keras_weights(base).items()}
    ctx = OB.Ctx(W, training=True)
    emb = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc)
    if not (len(trip) == 1 and tuple(trip[0]) == (p * k - 2, p * k - 1, 0)):       # (the reference's fallback triplet, :246-250)
        _semihard_sets_ok(trip, emb.detach().numpy(), p, k, margin)
    t = torch.as_tensor(trip, dtype=torch.long)
    rows = torch.clamp(((emb[t[:, 0]] - emb[t[:, 1]]) ** 2).sum(1) - ((emb[t[:, 0]] - emb[t[:, 2]]) ** 2).sum(1) + margin, min=0)
    assert len(trip) >= 1
    assert abs(mean.item() - rows.mean().item()) <= 1e-4 * abs(rows.mean().item()), (mean.item(), rows.mean().item())
    rows.mean().backward()
    got = B.keras_weights(base)
    num = den = 0.0
    for kk, pr in W.items():
        if pr.grad is None:
            continue
        diff = got[kk].grad.detach().cpu().double() - pr.grad
        num += (diff ** 2).sum().item()
        den += (pr.grad ** 2).sum().item()
    assert den > 0 and (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_shipped_configs_train_a_few_steps(dev):
    """Both configurations step with their own optimizer settings (RAdam, the YAML's learning rate): finite losses, finite
    weights, triplets of the right classes."""
    from embeddingnet_amd.train_step import TripletTrainer
    from embeddingnet_amd.utils import get_optimizer
    for which in ("b1", "b5"):
        cfg = _params(which)
        gen, trn, shape = cfg["generator"], cfg["train"], tuple(cfg["model"]["input_shape"])
        p, k = gen["k_classes"], gen["k_samples"]
        base, _ = _build(cfg, dev, seed=5)
        for mod in base.modules():
            if hasattr(mod, "enabled"):
                mod.enabled = True                   # the shipped behaviour: dropout and drop-connect active
        opt = get_optimizer(trn["optimizer"], trn["learning_rate"]).build([q for q in base.parameters() if q.requires_grad])
        tr = TripletTrainer(base, opt, p, k, margin=gen["margin"], negatives_selection_mode=gen["negatives_selection_mode"], seed=1)
        gent = torch.Generator(device=dev).manual_seed(6)
        proto = torch.rand((p,) + shape, device=dev, generator=gent)
        x = (proto.repeat_interleave(k, 0) + 0.2 * torch.randn((p * k,) + shape, device=dev, generator=gent)).clamp_(0, 1)
        hist = [tr.step(x).item() for _ in range(6)]
        assert all(np.isfinite(hist)), (which, hist)
        trip, count = tr.last_triplets
        t = trip[: int(count.item())].cpu().numpy()
        assert len(t) >= 1 and (t[:, 0] // k == t[:, 1] // k).all() and (t[:, 0] // k != t[:, 2] // k).all()
        for q in base.parameters():
            assert torch.isfinite(q).all()
