"""Direct float64-oracle assertions for the FUSED kernels of round 4.

tests/test_round4_gpu.py compares every fused kernel with the unfused HIP chain (a regression check: a slip in a helper
both forms share would pass it).  Here each fused form is compared, on one case each, with the float64 layer functions of
oracle/backbones.py (conv2d, batchnorm, depthwise_conv2d, maxpool, dense) composed exactly as the reference composes the
Keras layers (reference embedding_net/backbones.py:21-31,44-68,84-98), at 2e-5 of max|ref| like test_batchnorm_train —
and the kernel trace must show that the fused kernel is what ran:

  bn_bwd_apply_inrelu4 (+ dropout)   conv(+bias, ReLU) -> BatchNormalization [-> Dropout]
  maxpool_relu_bwd_colsum4           conv(+bias, ReLU) -> MaxPool2D
  bn_bwd_reduce4_gap / bn_bwd_gap    BatchNormalization(swish) with its pooled branch (squeeze-and-excite input)
  se_bn_sums4, affine_act_gap4 (y = NULL), affine_act_scale4      BatchNormalization.se_gate
  affine_drop_add4                   BatchNormalization -> DropConnect -> Add
  dwconv forward with the BN statistics, dwconv data gradient with the BN-backward sums
                                     BatchNormalization(swish) -> DepthwiseConv2D -> BatchNormalization(swish)
"""
import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib
from oracle import backbones as OB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def g(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def d64(a, grad=False):
    t = torch.tensor(np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a), dtype=torch.float64)
    return t.requires_grad_(True) if grad else t


def close(got, want, rtol, what):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = want.detach().double().numpy() if torch.is_tensor(want) else np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err <= rtol, f"{what}: max err / max|ref| = {err:.3e} > {rtol:.1e}"


class traced:
    """Kernel names launched inside the block (library trace)."""

    def __enter__(self):
        _lib.trace_reset(); _lib.trace_enable(True)
        self.names = []
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.names = [r[0] for r in _lib.trace_records()]
        _lib.trace_enable(False)
        return False

    def ran(self, part):
        return any(part in n for n in self.names)


def _bn(c, dev, act=None, relu=False, eps=1e-3):
    from embeddingnet_amd import layers as L
    bn = L.BatchNormalization(c, epsilon=eps, relu=relu, activation=act).to(dev).train()
    with torch.no_grad():
        bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
    return bn


def _bn_params(bn, name):
    c = bn.gamma.shape[0]
    return {f"{name}/gamma": d64(bn.gamma, True), f"{name}/beta": d64(bn.beta, True),
            f"{name}/moving_mean": torch.zeros(c, dtype=torch.float64), f"{name}/moving_variance": torch.ones(c, dtype=torch.float64)}


def _swish(t):
    return t * torch.sigmoid(t)


@pytest.mark.parametrize("with_dropout", [False, True])
def test_conv_relu_batchnorm_dropout_block_vs_oracle(dev, with_dropout):
    """simple2's block (reference backbones.py:44-55): Conv2D(bias, relu) -> BatchNormalization [-> Dropout(0.25)]."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(7)
    n, h, w, cin, cout, k, rate = 4, 17, 15, 16, 32, 3, 0.25
    x = rs.randn(n, h, w, cin).astype(np.float32)
    conv = L.Conv2D(cin, cout, k, activation="relu", gen=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        conv.bias.copy_(torch.linspace(-0.3, 0.3, cout))
    bn = _bn(cout, dev)
    drop = L.Dropout(rate, seed=11).train() if with_dropout else None
    xt = g(x, dev).requires_grad_(True)
    with traced() as tr:
        z = conv(xt, emit_stats=True)
        y = bn(z, dropout=drop) if with_dropout else bn(z)
        wgt = torch.cos(torch.arange(y.numel(), device=dev, dtype=torch.float32).reshape(y.shape) * 0.37)
        (y * wgt).sum().backward()
    assert tr.ran("bn_bwd_apply_inrelu4") and not tr.ran("relu_bwd"), tr.names
    if with_dropout:
        assert not tr.ran("dropout_kernel") and tr.ran("inrelu"), tr.names       # no Dropout pass of its own, forward or backward
    P = {"c/kernel": d64(conv.kernel, True), "c/bias": d64(conv.bias, True), **_bn_params(bn, "b")}
    ctx = OB.Ctx(P, training=True)
    xr = d64(x, True)
    yr = OB.batchnorm(ctx, "b", OB.conv2d(ctx, "c", xr, cout, k, relu=True))
    if with_dropout:
        # the device's mask (counter-based; the reference's is TensorFlow's RNG): where the output is exactly 0 although the
        # BatchNormalization's value is not.  Inverted scaling 1 / (1 - rate) on the survivors (Keras Dropout).
        keep = (y.detach().cpu().double() != 0) | (yr.detach().abs() < 1e-12)
        frac = 1.0 - keep.double().mean().item()
        assert abs(frac - rate) < 0.05, frac
        yr = yr * keep.double() / (1.0 - rate)
    close(y, yr, 1e-5, "block output")
    (yr * d64(wgt)).sum().backward()
    close(xt.grad, xr.grad, 2e-5, "dx")
    close(conv.kernel.grad, P["c/kernel"].grad, 2e-5, "dW")
    close(conv.bias.grad, P["c/bias"].grad, 2e-5, "dbias (summed inside the BatchNorm backward)")
    close(bn.gamma.grad, P["b/gamma"].grad, 2e-5, "dgamma")
    close(bn.beta.grad, P["b/beta"].grad, 2e-5, "dbeta")
    close(bn.moving_mean, ctx.new_stats["b/moving_mean"], 1e-5, "moving mean (conv-epilogue statistics)")
    close(bn.moving_variance, ctx.new_stats["b/moving_variance"], 1e-5, "moving variance")
    L.RELU_DONE.clear()


def test_conv_relu_maxpool_block_vs_oracle(dev):
    """`simple`'s block (reference backbones.py:21-31): Conv2D(bias, relu) -> MaxPool2D(): the pool's backward applies the ReLU
    mask and sums the bias gradient (embnet_maxpool_relu_bwd_colsum)."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(8)
    n, h, w, cin, cout, k = 3, 21, 18, 8, 64, 4
    x = rs.randn(n, h, w, cin).astype(np.float32)
    conv = L.Conv2D(cin, cout, k, activation="relu", gen=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        conv.bias.copy_(torch.linspace(-0.2, 0.4, cout))
    pool = L.MaxPool2D()
    xt = g(x, dev).requires_grad_(True)
    with traced() as tr:
        y = pool(conv(xt))
        wgt = torch.sin(torch.arange(y.numel(), device=dev, dtype=torch.float32).reshape(y.shape) * 0.61)
        (y * wgt).sum().backward()
    assert tr.ran("maxpool_relu_bwd_colsum4") and not tr.ran("relu_bwd_colsum_kernel"), tr.names
    P = {"c/kernel": d64(conv.kernel, True), "c/bias": d64(conv.bias, True)}
    ctx = OB.Ctx(P, training=True)
    xr = d64(x, True)
    yr = OB.maxpool(OB.conv2d(ctx, "c", xr, cout, k, relu=True))
    close(y, yr, 1e-5, "pooled output")
    (yr * d64(wgt)).sum().backward()
    close(xt.grad, xr.grad, 2e-5, "dx")
    close(conv.kernel.grad, P["c/kernel"].grad, 2e-5, "dW")
    close(conv.bias.grad, P["c/bias"].grad, 2e-5, "dbias (summed inside the pool backward)")
    L.RELU_DONE.clear()


def _se_reference(x, bn, se, P, ctx):
    """float64: a = swish(BN(x)); s = sigmoid(Dense(mean_hw a)); out = a * s  (an MBConv block's squeeze-and-excite)."""
    a = _swish(OB.batchnorm(ctx, "b", x))
    s = torch.sigmoid(a.mean(dim=(1, 2)) @ P["se/kernel"] + P["se/bias"])
    return a * s[:, None, None, :]


@pytest.mark.parametrize("form", ["emit_gap", "lazy_scale", "se_gate"])
def test_squeeze_excite_forms_vs_oracle(dev, form):
    """The three fused forms of BatchNormalization(swish) + squeeze-and-excite (reference backbones.py:84-98 via efficientnet's
    MBConv): emit_gap (pooled gradient inside the BN backward: bn_bwd_reduce4_gap), lazy_scale (+ the gate multiply's backward
    and the five per-(image, channel) sums: se_bn_sums4) and se_gate (the activated tensor is never written)."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(9)
    n, h, w, c = 5, 14, 9, 96
    x = (rs.randn(n, h, w, c) * 1.5 + 0.2).astype(np.float32)
    bn = _bn(c, dev, act="swish")
    se = L.Dense(c, c, gen=torch.Generator().manual_seed(2)).to(dev)
    xt = g(x, dev).requires_grad_(True)
    with traced() as tr:
        if form == "emit_gap":
            y, pooled = bn(xt, emit_gap=True)
            out = L.channel_scale(y, L.sigmoid(se(pooled)))
        elif form == "lazy_scale":
            y, pooled = bn(xt, emit_gap=True, lazy_scale=True)
            out = L.channel_scale(y, L.sigmoid(se(pooled)), lazy=True)
        else:
            out = bn.se_gate(xt, lambda pooled: L.sigmoid(se(pooled)))
        wgt = torch.cos(torch.arange(out.numel(), device=dev, dtype=torch.float32).reshape(out.shape) * 0.23)
        (out * wgt).sum().backward()
    if form == "emit_gap":
        assert tr.ran("bn_bwd_reduce4_gap") and not tr.ran("gap_bwd"), tr.names
    elif form == "lazy_scale":
        assert tr.ran("se_bn_sums4") and not tr.ran("bn_bwd_reduce"), tr.names
    else:
        assert tr.ran("se_bn_sums4") and tr.ran("affine_act_scale") and not tr.ran("chscale_fwd"), tr.names
    P = {**_bn_params(bn, "b"), "se/kernel": d64(se.kernel, True), "se/bias": d64(se.bias, True)}
    ctx = OB.Ctx(P, training=True)
    xr = d64(x, True)
    outr = _se_reference(xr, bn, se, P, ctx)
    close(out, outr, 1e-5, f"{form}: gated output")
    (outr * d64(wgt)).sum().backward()
    close(xt.grad, xr.grad, 2e-5, f"{form}: dx")
    close(bn.gamma.grad, P["b/gamma"].grad, 2e-5, f"{form}: dgamma")
    close(bn.beta.grad, P["b/beta"].grad, 2e-5, f"{form}: dbeta")
    close(se.kernel.grad, P["se/kernel"].grad, 2e-5, f"{form}: gate dense dW")
    close(se.bias.grad, P["se/bias"].grad, 2e-5, f"{form}: gate dense dbias")
    L.GATE_PENDING.clear()


@pytest.mark.parametrize("rate", [0.0, 0.3])
def test_batchnorm_dropconnect_add_vs_oracle(dev, rate):
    """An MBConv block's tail (project_bn -> DropConnect -> Add(block input)): BatchNormalization.drop_add, one forward pass
    (embnet_affine_drop_add), the per-sample drop factor applied inside the BatchNorm backward."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(10)
    n, h, w, c = 16, 7, 9, 80
    x, skip = rs.randn(n, h, w, c).astype(np.float32), rs.randn(n, h, w, c).astype(np.float32)
    bn = _bn(c, dev)
    drop = L.DropConnect(rate, seed=9).train() if rate > 0 else None
    xt, st = g(x, dev).requires_grad_(True), g(skip, dev).requires_grad_(True)
    with traced() as tr:
        out = bn.drop_add(xt, st, drop)
        wgt = torch.cos(torch.arange(out.numel(), device=dev, dtype=torch.float32).reshape(out.shape) * 0.41)
        (out * wgt).sum().backward()
    assert tr.ran("affine_drop_add4") and not tr.ran("sample_dropout"), tr.names
    P = _bn_params(bn, "b")
    ctx = OB.Ctx(P, training=True)
    xr, sr = d64(x, True), d64(skip, True)
    yr = OB.batchnorm(ctx, "b", xr)
    # the device's per-sample mask: a dropped sample's output IS the skip; survivors are scaled by 1 / (1 - rate)
    dropped = (out.detach().cpu() == torch.from_numpy(skip)).flatten(1).all(dim=1)
    if rate > 0:
        assert 0 < int(dropped.sum()) < n
    factor = (~dropped).double() / (1.0 - rate)
    outr = sr + yr * factor[:, None, None, None]
    close(out, outr, 1e-5, "drop_add output")
    (outr * d64(wgt)).sum().backward()
    close(xt.grad, xr.grad, 2e-5, "dx")
    close(st.grad, sr.grad, 2e-5, "dskip")
    close(bn.gamma.grad, P["b/gamma"].grad, 2e-5, "dgamma")
    close(bn.beta.grad, P["b/beta"].grad, 2e-5, "dbeta")


@pytest.mark.parametrize("k,stride", [(3, 1), (5, 2)])
def test_expand_bn_depthwise_bn_chain_vs_oracle(dev, k, stride):
    """An MBConv block's middle (expand_bn(swish) -> dwconv -> bn(swish)): the depthwise forward hands the second
    BatchNormalization its statistics, its data gradient hands the first one its backward sums."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(11)
    n, h, w, c = 4, 15, 17, 96
    x = rs.randn(n, h, w, c).astype(np.float32)
    bn1, bn2 = _bn(c, dev, act="swish"), _bn(c, dev, act="swish")
    dw = L.DepthwiseConv2D(c, k, strides=stride, gen=torch.Generator().manual_seed(4)).to(dev)
    xt = g(x, dev).requires_grad_(True)
    with traced() as tr:
        y = bn2(dw(bn1(xt), emit_stats=True))
        wgt = torch.cos(torch.arange(y.numel(), device=dev, dtype=torch.float32).reshape(y.shape) * 0.53)
        (y * wgt).sum().backward()
    assert not tr.ran("bn_stats4") or sum("bn_stats4" in nm for nm in tr.names) == 1, tr.names     # only bn1 computes its own
    assert sum("bn_bwd_reduce4" in nm for nm in tr.names) == 1, tr.names                         # only bn2 reduces; bn1 gets sums
    P = {**{kk.replace("b/", "b1/"): v for kk, v in _bn_params(bn1, "b").items()},
         **{kk.replace("b/", "b2/"): v for kk, v in _bn_params(bn2, "b").items()},
         "d/depthwise_kernel": d64(dw.depthwise_kernel, True)}
    ctx = OB.Ctx(P, training=True)
    xr = d64(x, True)
    yr = _swish(OB.batchnorm(ctx, "b2", OB.depthwise_conv2d(ctx, "d", _swish(OB.batchnorm(ctx, "b1", xr)), k, stride, OB.conv_normal)))
    close(y, yr, 1e-5, "chain output")
    (yr * d64(wgt)).sum().backward()
    close(xt.grad, xr.grad, 2e-5, "dx")
    close(dw.depthwise_kernel.grad, P["d/depthwise_kernel"].grad, 2e-5, "depthwise dW")
    for nm, bn in (("b1", bn1), ("b2", bn2)):
        close(bn.gamma.grad, P[f"{nm}/gamma"].grad, 2e-5, f"{nm} dgamma")
        close(bn.beta.grad, P[f"{nm}/beta"].grad, 2e-5, f"{nm} dbeta")
    close(bn2.moving_mean, ctx.new_stats["b2/moving_mean"], 1e-5, "bn2 moving mean (depthwise-epilogue statistics)")
    close(bn2.moving_variance, ctx.new_stats["b2/moving_variance"], 1e-5, "bn2 moving variance")
    L.BN_SUMS.clear()
