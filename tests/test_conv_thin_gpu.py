"""1x1 convolutions with a thin reduction as an HBM stream (csrc/conv_thin.hip: EfficientNet's expand convs forward, its project
convs' data gradient — reference embedding_net/backbones.py:84-98 via efficientnet's MBConv) against the float64 oracle layers:
output, the BatchNorm statistics taken from the kernel's own sums, input / kernel / bias gradients, a strided case, and the kernel
trace showing that the thin kernel is what ran."""
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


def close(got, want, rtol, what):
    got = got.detach().cpu().double().numpy()
    want = want.detach().double().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err <= rtol, f"{what}: max err / max|ref| = {err:.3e} > {rtol:.1e}"


def _names(fn):
    _lib.trace_reset(); _lib.trace_enable(True)
    out = fn()
    torch.cuda.synchronize()
    names = [r[0] for r in _lib.trace_records()]
    _lib.trace_enable(False)
    return out, names


@pytest.mark.parametrize("n,h,w,cin,cout", [(3, 17, 15, 16, 96), (2, 9, 11, 24, 144), (5, 7, 7, 40, 240), (4, 12, 10, 32, 16),
                                            (1, 1, 1, 8, 8), (2, 5, 5, 4, 1024)])
def test_expand_conv_with_batchnorm_statistics_vs_oracle(dev, n, h, w, cin, cout):
    """Conv2D(1x1, no bias) -> BatchNormalization(swish), the expand pair of an MBConv block: the thin kernel's output and the
    statistics it hands the BatchNormalization; then both layers' gradients (the conv's data gradient has a WIDE reduction here
    and stays on the MFMA kernel; its weight gradient too)."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(cin + cout)
    x = rs.randn(n, h, w, cin).astype(np.float32)
    conv = L.Conv2D(cin, cout, 1, use_bias=False, gen=torch.Generator().manual_seed(1)).to(dev)
    bn = L.BatchNormalization(cout, activation="swish").to(dev).train()
    with torch.no_grad():
        bn.gamma.copy_(torch.linspace(0.5, 1.5, cout)); bn.beta.copy_(torch.linspace(-0.3, 0.3, cout))
    xt = g(x, dev).requires_grad_(True)
    y, names = _names(lambda: bn(conv(xt, emit_stats=True)))
    thin = bool(_lib.lib().embnet_conv1x1_thin_supported(cin, cout))
    assert thin or (cin, cout) in ((32, 16), (8, 8)), (cin, cout)       # (an input tile beyond 4 float4 per thread stays on the MFMA kernel)
    assert any("thin_gemm" in nm for nm in names) == thin and not any("bn_stats" in nm for nm in names), names
    wgt = torch.cos(torch.arange(y.numel(), device=dev, dtype=torch.float32).reshape(y.shape) * 0.31)
    (y * wgt).sum().backward()
    P = {"c/kernel": conv.kernel.detach().cpu().double().requires_grad_(True),
         "b/gamma": bn.gamma.detach().cpu().double().requires_grad_(True), "b/beta": bn.beta.detach().cpu().double().requires_grad_(True),
         "b/moving_mean": torch.zeros(cout, dtype=torch.float64), "b/moving_variance": torch.ones(cout, dtype=torch.float64)}
    ctx = OB.Ctx(P, training=True)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    z = OB.batchnorm(ctx, "b", OB.conv2d(ctx, "c", xr, cout, 1, bias=False))
    yr = z * torch.sigmoid(z)
    close(y, yr, 1e-5, "conv -> bn output")
    close(bn.moving_mean, ctx.new_stats["b/moving_mean"], 1e-5, "moving mean from the thin kernel's sums")
    close(bn.moving_variance, ctx.new_stats["b/moving_variance"], 1e-5, "moving variance")
    (yr * wgt.cpu().double()).sum().backward()
    close(xt.grad, xr.grad, 2e-5, "dx")
    close(conv.kernel.grad, P["c/kernel"].grad, 2e-5, "dW")
    close(bn.gamma.grad, P["b/gamma"].grad, 2e-5, "dgamma")


@pytest.mark.parametrize("n,h,w,cin,cout", [(3, 17, 15, 96, 16), (2, 9, 11, 144, 24), (5, 7, 7, 240, 40), (4, 12, 10, 16, 8)])
def test_project_conv_data_gradient_vs_oracle(dev, n, h, w, cin, cout):
    """Conv2D(1x1) with many inputs and few outputs (an MBConv block's project conv): forward on the MFMA kernel, the DATA
    gradient — thin reduction, wide output — on the thin kernel."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(cin + cout)
    x = rs.randn(n, h, w, cin).astype(np.float32)
    dy = rs.randn(n, h, w, cout).astype(np.float32)
    conv = L.Conv2D(cin, cout, 1, use_bias=False, gen=torch.Generator().manual_seed(2)).to(dev)
    xt = g(x, dev).requires_grad_(True)
    y = conv(xt)
    _, names = _names(lambda: y.backward(g(dy, dev)))
    thin = bool(_lib.lib().embnet_conv1x1_thin_supported(cout, cin))
    assert thin and any("thin_gemm" in nm for nm in names), (names, thin)
    P = {"c/kernel": conv.kernel.detach().cpu().double().requires_grad_(True)}
    ctx = OB.Ctx(P, training=True)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr = OB.conv2d(ctx, "c", xr, cout, 1, bias=False)
    close(y, yr, 1e-5, "forward")
    yr.backward(torch.tensor(dy, dtype=torch.float64))
    close(xt.grad, xr.grad, 1e-5, "dx (thin kernel)")
    close(conv.kernel.grad, P["c/kernel"].grad, 2e-5, "dW")


def test_thin_conv_with_bias_relu_and_stride(dev):
    """The same kernel behind a biased, ReLU'd, stride-2 1x1 Conv2D ('valid'): output pixel (oh, ow) reads input (2 oh, 2 ow)."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(5)
    n, h, w, cin, cout = 3, 13, 10, 12, 40
    x = rs.randn(n, h, w, cin).astype(np.float32)
    conv = L.Conv2D(cin, cout, 1, strides=2, activation="relu", gen=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        conv.bias.copy_(torch.linspace(-0.5, 0.5, cout))
    xt = g(x, dev).requires_grad_(True)
    y, names = _names(lambda: conv(xt))
    assert _lib.lib().embnet_conv1x1_thin_supported(cin, cout) == 1 and any("thin_gemm" in nm for nm in names), names
    P = {"c/kernel": conv.kernel.detach().cpu().double().requires_grad_(True), "c/bias": conv.bias.detach().cpu().double().requires_grad_(True)}
    ctx = OB.Ctx(P, training=True)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr = OB.conv2d(ctx, "c", xr, cout, 1, stride=2, relu=True)
    assert tuple(y.shape) == tuple(yr.shape) == (n, 7, 5, cout)
    close(y, yr, 1e-5, "strided thin conv")
    wgt = rs.randn(*yr.shape)
    (y * g(wgt, dev)).sum().backward()
    (yr * torch.tensor(wgt)).sum().backward()
    close(xt.grad, xr.grad, 2e-5, "dx")
    close(conv.kernel.grad, P["c/kernel"].grad, 2e-5, "dW")
    close(conv.bias.grad, P["c/bias"].grad, 2e-5, "dbias")


@pytest.mark.parametrize("n,h,w,cin,cout,stride", [(6, 17, 15, 16, 96, 1), (4, 9, 11, 144, 24, 1), (5, 7, 7, 40, 240, 1), (3, 12, 10, 32, 16, 1),
                                                   (130, 14, 14, 24, 144, 1), (2, 13, 9, 12, 40, 2), (2, 6, 6, 20, 8, 1)])
def test_thin_weight_gradient_vs_oracle(dev, n, h, w, cin, cout, stride):
    """dW of a 1x1 conv with a thin side (thin_wgrad_kernel): x thin (expand), dy thin (project: the transposed slab), a strided
    case, many workgroups (130 images) and few; the trace names the kernel; two launches give bit-identical gradients."""
    from embeddingnet_amd import layers as L
    rs = np.random.RandomState(cin * 7 + cout)
    x = rs.randn(n, h, w, cin).astype(np.float32)
    conv = L.Conv2D(cin, cout, 1, strides=stride, use_bias=False, gen=torch.Generator().manual_seed(4)).to(dev)
    xt = g(x, dev).requires_grad_(True)
    y = conv(xt)
    dy = rs.randn(*y.shape).astype(np.float32)
    _, names = _names(lambda: y.backward(g(dy, dev)))
    assert any("thin_wgrad" in nm for nm in names) and not any("conv_wgrad_kernel" in nm for nm in names), names
    P = {"c/kernel": conv.kernel.detach().cpu().double().requires_grad_(True)}
    ctx = OB.Ctx(P, training=True)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr = OB.conv2d(ctx, "c", xr, cout, 1, stride=stride, bias=False)
    yr.backward(torch.tensor(dy, dtype=torch.float64))
    close(conv.kernel.grad, P["c/kernel"].grad, 2e-5, "dW (thin kernel)")
    close(xt.grad, xr.grad, 2e-5, "dx")
    first = conv.kernel.grad.clone()
    conv.kernel.grad = None
    xt.grad = None
    conv(xt).backward(g(dy, dev))
    assert torch.equal(conv.kernel.grad, first)
