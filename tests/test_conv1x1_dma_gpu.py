"""1x1 convolutions with the activation operand read as fp32 by LDS-DMA (csrc/conv_patch.hip conv1x1_a32_kernel,
embnet_conv2d_dma1x1_f32; ABI 22): the zoo ResNets' bottleneck conv1 / conv3 and projection shortcuts (reference
embedding_net/backbones.py:99-104) without the gather loop and without planes of the activation.  Through the C ABI, against float64:

  * forward at the ResNet50 channel classes, ragged pixel counts, K not a multiple of the tile, stride 2, few tiles (the reduction
    split over workgroups + fix-up), activation amplitudes 1e-4 ... 3e4 with an 8x-loose range (the operand's scale comes from its
    range slot);
  * integer operands exact (any row / quad / stage slip of the swizzled DMA layout shows as an integer error);
  * the epilogue options (bias + ReLU, residual, BatchNorm statistics by row band);
  * the stride-1 data gradient (dy x flip-1 kernel planes), plain and with a gradient to add;
  * bench sizes (batch 256) against the three-product gather kernel, bit-for-bit repeatable;
  * in the network: a bottleneck unit with layers.CONV1X1_DMA on / off agrees to fp32 rounding, its trace shows the kernel.
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
        pytest.skip("built for the two-piece fp16 planes format")
    return torch.device("cuda", 0)


def slot_for(t, loose=1.0):
    """A range slot holding loose * max |t| (an upper bound, as a BatchNormalization's statistics give)."""
    return (t.detach().abs().max() * loose).reshape(1).float().view(torch.int32).clone()


def conv1x1(x, w, stride=1, bias=None, relu=0, residual=None, stats=False, flip=0, loose=1.0):
    """flip = 1: x is dy [n,h,w,k], w the [1,1,c,k] kernel -> dx [n,h,w,c]."""
    lib = _lib.lib()
    n, h, wd, cin = x.shape
    c, k = w.shape[2], w.shape[3]
    red, cols = (k, c) if flip else (c, k)
    assert cin == red
    oh, ow = (h - 1) // stride + 1, (wd - 1) // stride + 1
    assert lib.embnet_conv2d_dma1x1_supported(n, h, wd, red, cols, stride, oh, ow) == 1
    y = torch.full((n, oh, ow, cols), float("nan"), device=x.device)
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(n, red, 1, 1, cols, oh, ow), 4) // 4, device=x.device)
    rows = lib.embnet_conv2d_patch_stats_rows(n, oh, ow)
    st = torch.full((2, cols, rows), float("nan"), device=x.device) if stats else None
    rng = slot_for(x, loose)
    _lib.check(lib.embnet_conv2d_dma1x1_f32(x.data_ptr(), L.weight_planes(w, flip).data_ptr(), _lib.ptr(bias), y.data_ptr(), n, h, wd, red,
                                            cols, stride, oh, ow, relu, _lib.ptr(residual), _lib.ptr(st), rng.data_ptr(), ws.data_ptr(),
                                            ws.numel() * 4, _lib.stream()))
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


@pytest.mark.parametrize("amplitude", [1e-4, 1e-2, 1.7, 3e4])
def test_activation_amplitudes_with_a_loose_range(dev, amplitude):
    rng = np.random.default_rng(5)
    x = (np.maximum(rng.standard_normal((6, 14, 14, 256)), 0) * amplitude).astype(np.float32)
    w = (rng.standard_normal((1, 1, 256, 512)) / 16).astype(np.float32)
    y = conv1x1(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), loose=8.0)
    want = ref64(x, w)
    err = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 1.5e-6, err


def test_integer_operands_are_exact(dev):
    torch.manual_seed(3)
    for (n, h, c, k, stride) in ((3, 9, 64, 128, 1), (2, 11, 128, 64, 2), (5, 7, 256, 192, 1), (2, 33, 32, 256, 1)):
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
    yv = y.cpu().numpy().astype(np.float64).reshape(-1, k)
    s1, s2 = st.cpu().numpy().astype(np.float64).sum(axis=2)
    assert np.isfinite(st.cpu().numpy()).all()
    assert np.abs(s1 - yv.sum(0)).max() <= 1e-4 * np.abs(yv).sum(0).max()
    assert np.abs(s2 - (yv ** 2).sum(0)).max() <= 1e-5 * (yv ** 2).sum(0).max()
    assert (st[1].amax(dim=1).cpu().numpy() >= (yv ** 2).max(0) * (1 - 1e-6)).all()


@pytest.mark.parametrize("n,h,c,k", [(4, 28, 128, 512), (6, 14, 1024, 256), (3, 13, 64, 96), (8, 7, 2048, 512)])
def test_data_gradient_vs_float64(dev, n, h, c, k):
    rng = np.random.default_rng(n * h + k)
    dy = (rng.standard_normal((n, h, h, k)) * 1e-5).astype(np.float32)
    w = (rng.standard_normal((1, 1, c, k)) / np.sqrt(c)).astype(np.float32)
    add = (rng.standard_normal((n, h, h, c)) * 1e-5).astype(np.float32)
    want = np.einsum("nhwk,ck->nhwc", dy.astype(np.float64), w[0, 0].astype(np.float64))
    dx = conv1x1(torch.from_numpy(dy).to(dev), torch.from_numpy(w).to(dev), flip=1)
    assert np.abs(dx.cpu().numpy() - want).max() / np.abs(want).max() < 1.5e-6
    dx = conv1x1(torch.from_numpy(dy).to(dev), torch.from_numpy(w).to(dev), flip=1, residual=torch.from_numpy(add).to(dev))
    assert np.abs(dx.cpu().numpy() - (want + add)).max() / np.abs(want + add).max() < 1.5e-6


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


def test_the_range_slot_is_required(dev):
    lib = _lib.lib()
    x = torch.rand((2, 8, 8, 64), device=dev)
    w = torch.rand((1, 1, 64, 128), device=dev)
    y = torch.empty((2, 8, 8, 128), device=dev)
    rc = lib.embnet_conv2d_dma1x1_f32(x.data_ptr(), L.weight_planes(w, 0).data_ptr(), None, y.data_ptr(), 2, 8, 8, 64, 128, 1, 8, 8, 0, None, None,
                                      None, None, 0, _lib.stream())
    assert rc != 0 and b"range slot" in lib.embnet_last_error()
    assert lib.embnet_conv2d_dma1x1_supported(2, 8, 8, 48, 128, 1, 8, 8) == 0          # c % 32 != 0


def test_bottleneck_unit_on_the_dma_kernel(dev):
    """layers.CONV1X1_DMA on (it is off by default): the unit's 1x1 convs with >= 128 output channels run their forward — and the
    stride-1 data gradients that carry no BatchNorm sums — on conv1x1_a32_kernel (trace); outputs and every gradient agree with the
    gather kernels to fp32 rounding; the BatchNorm backward launches no reduce pass more than with the gather kernels."""
    from embeddingnet_amd.backbones import ResidualUnit

    def run(on):
        old = L.CONV1X1_DMA[0]
        L.CONV1X1_DMA[0] = on
        try:
            unit = ResidualUnit(1024, 256, 1, False, "bottleneck", torch.Generator().manual_seed(5)).to(dev).train()
            g = torch.Generator(device=dev).manual_seed(6)
            x = torch.randn((8, 14, 14, 1024), device=dev, generator=g).requires_grad_(True)
            x._range = slot_for(x, 2.0)                  # (in the network: the BatchNormalization in front vouches for it)
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
            L.CONV1X1_DMA[0] = old

    y1, dx1, g1, n1 = run(1)
    y0, dx0, g0, n0 = run(0)
    assert not any("conv1x1_a32_kernel" in s for s in n0)
    a32 = [s for s in n1 if "conv1x1_a32_kernel" in s]
    assert len(a32) >= 2, (a32, [s for s in n1 if "conv_" in s])          # conv1 (1024 -> 256) and conv3 (256 -> 1024) forward at least
    assert sum("bn_bwd_reduce" in s for s in n1) == sum("bn_bwd_reduce" in s for s in n0)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y1, y0) < 2e-6 and rel(dx1, dx0) < 2e-5
    for a, b in zip(g1, g0):
        assert rel(a, b) < 2e-5
