"""1x1 convolutions on the pre-split planes (csrc/conv_patch.hip conv1x1_planes_kernel, embnet_conv2d_planes1x1_f32; VERDICT r03-r05
#2a): the bottleneck units' conv1 / conv3 and the projection shortcuts of the zoo ResNets (reference embedding_net/backbones.py:99-104)
as a GEMM fed by LDS-DMA.  Through the C ABI, against float64:

  * forward at the ResNet50 channel classes, ragged pixel counts, K not a multiple of the tile, stride 2 (shortcuts), few tiles
    (the reduction split over workgroups + fix-up);
  * the epilogue options (bias + ReLU, residual, BatchNorm statistics by row band);
  * the stride-1 data gradient (dy planes x flip-1 kernel planes);
  * bench sizes (batch 256) against the three-product gather kernel, and bit-for-bit repeatable.
"""
import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib
from embeddingnet_amd import layers as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    if _lib.lib().embnet_conv_planes_mfma_terms() != 3:
        pytest.skip("the 1x1 planes kernel is built for the two-piece fp16 planes format")
    return torch.device("cuda", 0)


def planes_of(x):
    c = x.shape[-1]
    p = torch.empty(3 * x.numel(), device=x.device, dtype=torch.int16)
    _lib.check(_lib.lib().embnet_planes_from_f32(x.data_ptr(), x.numel() // c, c, p.data_ptr(), _lib.stream()))
    return p


def conv1x1(x, w, stride=1, bias=None, relu=0, residual=None, stats=False, flip=0):
    """y = planes GEMM; flip = 1: x is dy [n,h,w,k], w the [1,1,c,k] kernel -> dx [n,h,w,c]."""
    lib = _lib.lib()
    n, h, wd, cin = x.shape
    c, k = w.shape[2], w.shape[3]
    red, cols = (k, c) if flip else (c, k)
    assert cin == red
    oh, ow = (h - 1) // stride + 1, (wd - 1) // stride + 1
    assert lib.embnet_conv2d_patch_supported(n, red, 1, 1, cols, stride, oh, ow) == 1
    y = torch.full((n, oh, ow, cols), float("nan"), device=x.device)
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(n, red, 1, 1, cols, oh, ow), 4) // 4, device=x.device)
    rows = lib.embnet_conv2d_patch_stats_rows(n, oh, ow)
    st = torch.full((2, cols, rows), float("nan"), device=x.device) if stats else None
    xp = planes_of(x)
    _lib.check(lib.embnet_conv2d_planes1x1_f32(xp.data_ptr(), L.weight_planes(w, flip).data_ptr(), _lib.ptr(bias), y.data_ptr(), n, h, wd, red,
                                               cols, stride, oh, ow, relu, _lib.ptr(residual), _lib.ptr(st), ws.data_ptr(), ws.numel() * 4,
                                               _lib.stream()))
    return (y, st) if stats else y


def ref64(x, w, stride=1):
    return np.einsum("nhwc,ck->nhwk", x[:, ::stride, ::stride].astype(np.float64), w[0, 0].astype(np.float64))


GEOMS = [  # n, h, w, c, k, stride
    (4, 56, 56, 64, 256, 1), (4, 56, 56, 256, 64, 1), (4, 28, 28, 128, 512, 1), (6, 14, 14, 1024, 256, 1), (8, 7, 7, 512, 2048, 1),
    (8, 7, 7, 2048, 512, 1),      # few tiles, long reduction: split over workgroups + fix-up
    (4, 56, 56, 256, 512, 2), (3, 15, 13, 64, 96, 2), (3, 13, 9, 32, 100, 1), (1, 5, 5, 64, 64, 1), (2, 20, 31, 96, 160, 1),
]


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_forward_vs_float64(dev, geom):
    n, h, wd, c, k, stride = geom
    rng = np.random.default_rng(sum(geom))
    x = (np.maximum(rng.standard_normal((n, h, wd, c)), 0) * 0.03).astype(np.float32)
    w = (rng.standard_normal((1, 1, c, k)) / np.sqrt(c)).astype(np.float32)
    y = conv1x1(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), stride)
    want = ref64(x, w, stride)
    err = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    assert np.isfinite(y.cpu().numpy()).all() and err < 1.5e-6, err


def test_integer_operands_are_exact(dev):
    """Every product and sum exact: any row / chunk / stage slip shows as an integer error."""
    torch.manual_seed(3)
    for (n, h, c, k, stride) in ((3, 9, 64, 128, 1), (2, 11, 128, 64, 2), (5, 7, 256, 192, 1)):
        x = torch.randint(-3, 4, (n, h, h, c), device=dev).float()
        w = torch.randint(-2, 3, (1, 1, c, k), device=dev).float()
        y = conv1x1(x, w, stride)
        want = ref64(x.cpu().numpy(), w.cpu().numpy(), stride)
        assert np.array_equal(y.cpu().numpy().astype(np.float64), want)


@pytest.mark.parametrize("n,h,c,k", [(8, 28, 128, 512), (5, 11, 64, 96), (32, 7, 512, 2048)])
def test_epilogues(dev, n, h, c, k):
    rng = np.random.default_rng(n + h + c)
    x = torch.from_numpy(rng.standard_normal((n, h, h, c)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((1, 1, c, k)) / np.sqrt(c)).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.standard_normal(k).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.standard_normal((n, h, h, k)).astype(np.float32)).to(dev)
    base = ref64(x.cpu().numpy(), w.cpu().numpy())
    y = conv1x1(x, w, bias=bias, relu=1)
    want = np.maximum(base + bias.cpu().numpy(), 0)
    assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < 2e-6
    y, st = conv1x1(x, w, residual=res, stats=True)
    want = base + res.cpu().numpy()
    assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < 2e-6
    # statistics of the values written (the Add included), by 64-row band: their sums are the column sums / sums of squares
    yv = y.cpu().numpy().astype(np.float64).reshape(-1, k)
    s1, s2 = st.cpu().numpy().astype(np.float64).sum(axis=2)
    assert np.isfinite(st.cpu().numpy()).all()
    assert np.abs(s1 - yv.sum(0)).max() <= 1e-4 * np.abs(yv).sum(0).max()
    assert np.abs(s2 - (yv ** 2).sum(0)).max() <= 1e-5 * (yv ** 2).sum(0).max()
    # ... and every element's square is a term of exactly one band's sum (what the output bound of the BatchNorm behind relies on)
    assert (st[1].amax(dim=1).cpu().numpy() >= (yv ** 2).max(0) * (1 - 1e-6)).all()


@pytest.mark.parametrize("n,h,c,k", [(4, 28, 128, 512), (6, 14, 1024, 256), (3, 13, 64, 96)])
def test_data_gradient_vs_float64(dev, n, h, c, k):
    if k % 16:
        pytest.skip("the flip-1 kernel planes need k % 16 == 0")
    rng = np.random.default_rng(n * h + k)
    dy = (rng.standard_normal((n, h, h, k)) * 1e-5).astype(np.float32)
    w = (rng.standard_normal((1, 1, c, k)) / np.sqrt(c)).astype(np.float32)
    dx = conv1x1(torch.from_numpy(dy).to(dev), torch.from_numpy(w).to(dev), flip=1)
    want = np.einsum("nhwk,ck->nhwc", dy.astype(np.float64), w[0, 0].astype(np.float64))
    err = np.abs(dx.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 1.5e-6, err


@pytest.mark.parametrize("n,h,c,k,stride", [(256, 56, 64, 256, 1), (256, 28, 128, 512, 1), (256, 14, 256, 1024, 1), (256, 7, 512, 2048, 1),
                                             (256, 14, 1024, 256, 1), (256, 56, 256, 512, 2)])
def test_bench_sizes_vs_gather_kernel_and_repeatable(dev, n, h, c, k, stride):
    lib = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(n + h + c)
    x = torch.relu(torch.randn((n, h, h, c), device=dev, generator=g))
    w = torch.randn((1, 1, c, k), device=dev, generator=g) * (2.0 / c) ** 0.5
    y1 = conv1x1(x, w, stride)
    y2 = conv1x1(x, w, stride)
    assert torch.equal(y1, y2)
    oh = (h - 1) // stride + 1
    y6 = torch.empty((n, oh, oh, k), device=dev)
    ws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, c, 1, 1, k, oh, oh) // 4, 4), device=dev)
    _lib.check(lib.embnet_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), None, y6.data_ptr(), n, h, h, c, 1, 1, k, stride, 0, 0, oh, oh, 0, None, None,
                                         None, 0, None, ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    err = float((y1 - y6).abs().max() / y6.abs().max())
    assert err < 2e-6, err


def test_bottleneck_unit_with_its_1x1_convs_on_the_planes_gemm(dev):
    """layers.CONV1X1_PLANES on: bn1 / bn3 write their outputs also as planes and conv1 / conv3 run their forward on the planes GEMM (trace); outputs and
    every gradient agree with the default wiring (gather kernels) to fp32 rounding; nothing is left in the step context."""
    from embeddingnet_amd.backbones import ResidualUnit

    def run(on):
        old = L.CONV1X1_PLANES[0]
        L.CONV1X1_PLANES[0] = on
        try:
            unit = ResidualUnit(1024, 256, 1, False, "bottleneck", torch.Generator().manual_seed(5)).to(dev).train()   # conv1 AND conv3 marked
            g = torch.Generator(device=dev).manual_seed(6)
            x = torch.randn((8, 14, 14, 1024), device=dev, generator=g).requires_grad_(True)
            dy = torch.randn((8, 14, 14, 1024), device=dev, generator=g) * 1e-3
            _lib.trace_reset(); _lib.trace_enable(True)
            try:
                y = unit(x)
                y.backward(dy)
                names = [r[0] for r in _lib.trace_records()]
            finally:
                _lib.trace_enable(False)
            assert not L.current_context().leftovers()
            return y.detach(), x.grad, [p.grad.clone() for p in unit.parameters()], names
        finally:
            L.CONV1X1_PLANES[0] = old

    y1, dx1, g1, n1 = run(True)
    y0, dx0, g0, n0 = run(False)
    assert sum("conv1x1_planes_kernel" in s for s in n1) == 2 and not any("conv1x1_planes_kernel" in s for s in n0), (n1, n0)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y1, y0) < 2e-6 and rel(dx1, dx0) < 2e-5
    for a, b in zip(g1, g0):
        assert rel(a, b) < 2e-5
