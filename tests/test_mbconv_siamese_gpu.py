"""GPU parity: MBConv kernels, EfficientNet-B0, and the siamese (contrastive) path vs the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import backbones as OB
from oracle import losses as olosses

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def g(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def close(got, want, rtol, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = want.detach().double().numpy() if torch.is_tensor(want) else np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err <= rtol, f"{what}: max err / max|ref| = {err:.3e} > {rtol:.1e}"


@pytest.mark.parametrize("n,h,w,c,k,s", [(2, 16, 16, 32, 3, 1), (3, 15, 17, 96, 3, 2), (2, 14, 14, 240, 5, 2),
                                         (2, 7, 7, 672, 5, 1), (2, 9, 9, 6, 3, 1),
                                         # register-blocked kernels: even and odd left padding at stride 2, ragged column blocks
                                         (2, 16, 16, 32, 3, 2), (2, 15, 15, 32, 5, 2), (1, 13, 18, 16, 5, 1), (1, 12, 22, 40, 3, 1),
                                         (2, 21, 10, 24, 5, 2), (3, 10, 6, 8, 3, 2)])
def test_depthwise_conv(dev, n, h, w, c, k, s):
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(n * 100 + c)
    layer = L.DepthwiseConv2D(c, k, s).to(dev)
    x = rs.randn(n, h, w, c).astype(np.float32)
    kern = rs.randn(k, k, c, 1).astype(np.float32) * 0.3
    with torch.no_grad():
        layer.depthwise_kernel.copy_(g(kern, dev))
    xt = g(x, dev).requires_grad_(True)
    y = layer(xt)
    ctx = OB.Ctx({"d/depthwise_kernel": torch.tensor(kern, dtype=torch.float64, requires_grad=True)})
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr = OB.depthwise_conv2d(ctx, "d", xr, k, s, OB.conv_normal)
    close(y, yr, 1e-5, "dw fwd")
    dy = rs.randn(*yr.shape).astype(np.float32)
    y.backward(g(dy, dev))
    yr.backward(torch.tensor(dy, dtype=torch.float64))
    close(xt.grad, xr.grad, 1e-5, "dw dgrad")
    close(layer.depthwise_kernel.grad, ctx.params["d/depthwise_kernel"].grad, 2e-5, "dw wgrad")


def test_activations_channel_scale_absdiff(dev):
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(1)
    x = (rs.randn(3, 5, 7, 24) * 3).astype(np.float32)
    for fn, ref in ((L.sigmoid, torch.sigmoid), (L.swish, lambda t: t * torch.sigmoid(t))):
        xt = g(x, dev).requires_grad_(True)
        xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        y, yr = fn(xt), ref(xr)
        close(y, yr, 2e-6, "act fwd")
        w = rs.randn(*x.shape)
        (y * g(w, dev)).sum().backward()
        (yr * torch.tensor(w)).sum().backward()
        close(xt.grad, xr.grad, 1e-5, "act bwd")
    s = rs.rand(3, 24).astype(np.float32)
    xt, st = g(x, dev).requires_grad_(True), g(s, dev).requires_grad_(True)
    xr, sr = torch.tensor(x, dtype=torch.float64, requires_grad=True), torch.tensor(s, dtype=torch.float64, requires_grad=True)
    y = L.channel_scale(xt, st)
    yr = xr * sr[:, None, None, :]
    close(y, yr, 1e-6, "se scale")
    w = rs.randn(*x.shape)
    (y * g(w, dev)).sum().backward()
    (yr * torch.tensor(w)).sum().backward()
    close(xt.grad, xr.grad, 1e-6, "se dx")
    close(st.grad, sr.grad, 1e-5, "se ds")
    a, b = rs.randn(9, 40).astype(np.float32), rs.randn(9, 40).astype(np.float32)
    at, bt = g(a, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
    d = L.abs_diff(at, bt)
    assert np.array_equal(d.detach().cpu().numpy(), np.abs(a - b))
    d.sum().backward()
    assert np.array_equal(at.grad.cpu().numpy(), np.sign(a - b)) and np.array_equal(bt.grad.cpu().numpy(), -np.sign(a - b))
    # drop-connect: whole samples dropped, survivors scaled, same mask in backward
    dc = L.DropConnect(0.5, seed=3).to(dev).train()
    xt = torch.ones(64, 2, 2, 4, device=dev, requires_grad=True)
    y = dc(xt)
    per = y.detach().reshape(64, -1)
    assert torch.all((per == 0).all(1) | (per == 2).all(1)) and 10 < int((per[:, 0] == 0).sum()) < 54
    y.sum().backward()
    assert torch.equal(xt.grad, y.detach())


def _oracle_from(model, training, dtype=torch.float64):
    from embeddingnet_amd.backbones import keras_weights
    params = {k: v.detach().cpu().to(dtype).requires_grad_(v.requires_grad) for k, v in keras_weights(model).items()}
    return OB.Ctx(params, training=training)


def test_efficientnet_b0_vs_oracle(dev):
    from embeddingnet_amd import backbones as B
    name, shape, enc, batch = "efficientnet-b0", (96, 96, 3), 64, 6
    base, backbone = B.get_backbone(shape, encodings_len=enc, backbone_name=name, backbone_weights=None, seed=1, device=dev)
    assert sum(p.numel() for p in backbone.parameters()) == pytest.approx(4.01e6, rel=0.02)   # B0 without top
    for m in base.modules():
        if hasattr(m, "enabled"):
            m.enabled = False                    # drop-connect off for parity
    rs = np.random.RandomState(0)
    x = rs.rand(batch, *shape).astype(np.float32)
    wgt = rs.randn(batch, enc).astype(np.float32)
    base.train()
    emb = base(g(x, dev))
    (emb * g(wgt, dev)).sum().backward()
    ctx = _oracle_from(base, True)
    embr = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc)
    close(emb, embr, 5e-4, "effnet embeddings")
    (embr * torch.tensor(wgt, dtype=torch.float64)).sum().backward()
    ctx32 = _oracle_from(base, True, torch.float32)
    emb32 = OB.base_model(ctx32, torch.tensor(x), backbone_name=name, encodings_len=enc)
    (emb32 * torch.tensor(wgt)).sum().backward()
    got = B.keras_weights(base)
    bad, num, den, total = [], 0.0, 0.0, 0
    for k, p in ctx.params.items():
        if p.grad is None:
            continue
        total += 1
        diff = got[k].grad.detach().cpu().double() - p.grad
        scale = max(p.grad.abs().max().item(), 1e-12)
        err = diff.abs().max().item() / scale
        floor = (ctx32.params[k].grad.double() - p.grad).abs().max().item() / scale
        num += (diff ** 2).sum().item()
        den += (p.grad ** 2).sum().item()
        if err >= 5 * floor + 1e-4:
            bad.append(f"{k}: {err:.2e} (floor {floor:.2e})")
    assert len(bad) <= total // 40, bad
    assert (num / den) ** 0.5 < 2e-2
    pred = base.predict(x)
    ctx_i = _oracle_from(base, False)
    close(pred, OB.base_model(ctx_i, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc), 5e-4,
          "effnet predict")


@pytest.mark.parametrize("distance_type", ["l2", "l1"])
def test_siamese_contrastive_step_vs_oracle(dev, distance_type):
    """C3 in miniature: SiameseNet (simple2 backbone) + contrastive_loss + accuracy, loss and gradients."""
    from embeddingnet_amd.backbones import keras_weights
    from embeddingnet_amd.losses_and_accuracies import accuracy, contrastive_loss
    from embeddingnet_amd.models import SiameseNet
    enc, b = 32, 8
    params = {"model": dict(input_shape=[64, 64, 3], encodings_len=enc, mode="siamese", distance_type=distance_type,
                            backbone_name="simple2", backbone_weights=None, freeze_backbone=False,
                            embeddings_normalization=True, device=dev, seed=4),
              "dataloader": {}, "generator": {}, "train": {}, "general": {"work_dir": "w/", "project_name": "p"}}
    net = SiameseNet(params, training=True)
    for m in net.model.modules():
        if hasattr(m, "enabled"):
            m.enabled = False
    rs = np.random.RandomState(2)
    x1, x2 = rs.rand(b, 64, 64, 3).astype(np.float32), rs.rand(b, 64, 64, 3).astype(np.float32)
    y = np.zeros((b, 1), np.float32)
    y[: b // 2] = 1                                       # generator layout: first half same-class
    net.model.train()
    out = net.model([g(x1, dev), g(x2, dev)])[0]
    loss = contrastive_loss(g(y, dev), out)
    acc = accuracy(g(y, dev), out)
    loss.backward()
    W = {k: v.detach().cpu().double().requires_grad_(v.requires_grad) for k, v in keras_weights(net.model).items()}
    ctx = OB.Ctx(W, training=True)
    kw = dict(backbone_name="simple2", encodings_len=enc)
    e1 = OB.base_model(ctx, torch.tensor(x1, dtype=torch.float64), **kw)
    e2 = OB.base_model(ctx, torch.tensor(x2, dtype=torch.float64), **kw)
    if distance_type == "l2":
        d = OB.siamese_l2_distance(e1, e2)
    else:
        d = torch.sigmoid(torch.abs(e1 - e2) @ W["output_siamese/kernel"] + W["output_siamese/bias"])
    yt = torch.tensor(y, dtype=torch.float64)
    lr = (yt * d ** 2 + (1 - yt) * torch.clamp(1 - d, min=0) ** 2).mean()
    assert abs(lr.item() - olosses.contrastive_loss(y, d.detach().numpy())) < 1e-12
    assert abs(loss.item() - lr.item()) <= 1e-4 * abs(lr.item())
    assert acc.item() == pytest.approx(olosses.accuracy(y, d.detach().numpy()))
    lr.backward()
    got = keras_weights(net.model)
    for k, p in W.items():
        if p.grad is None:
            continue
        err = (got[k].grad.detach().cpu().double() - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-12)
        assert err < 2e-3, f"{distance_type}: grad {k} rel err {err:.2e}"
