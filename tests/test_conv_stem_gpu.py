"""The zoo ResNets' stem convolution (7x7, stride 2, 3 -> 64 channels on the image; reference embedding_net/backbones.py:99-104 via
image-classifiers: bn_data -> ZeroPadding2D(3) -> conv0) forward on its own kernel (csrc/conv_stem.hip, embnet_conv2d_stem_f32,
ABI 22).  Through the C ABI, against float64:

  * image sizes 224 / 64 / ragged (outputs not a multiple of the 16 x 16 tile; odd sizes), paddings 3 and 0, more tiles than
    workgroups (the persistent loop and its double-buffered patch), amplitudes 1e-3 ... 1e3 with loose ranges;
  * integer operands exact (any patch row / column-flip / tap-order slip is an integer error);
  * the statistics partials: their sums are the output's column sums / sums of squares, every square inside one band;
  * at the bench size (batch 128 x 224 x 224) against the three-product gather kernel, bit-for-bit repeatable;
  * in the network: ResNet18's trace shows the kernel, and outputs / gradients equal the gather kernel's to fp32 rounding.
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
        pytest.skip("built for the two-piece fp16 format")
    return torch.device("cuda", 0)


def slot_for(t, loose=1.0):
    return (t.detach().abs().max() * loose).reshape(1).float().view(torch.int32).clone()


def stem(x, w, pad, stats=False, loose=(1.0, 1.0)):
    lib = _lib.lib()
    n, h, wd, c = x.shape
    oh, ow = (h + 2 * pad - 7) // 2 + 1, (wd + 2 * pad - 7) // 2 + 1
    assert c == 4 and tuple(w.shape) == (7, 7, 4, 64)
    assert lib.embnet_conv2d_stem_supported(n, h, wd, 4, 7, 7, 64, 2, pad, pad, oh, ow) == 1
    y = torch.full((n, oh, ow, 64), float("nan"), device=x.device)
    rows = lib.embnet_conv2d_stem_stats_rows(n, oh, ow)
    st = torch.full((2, 64, rows), float("nan"), device=x.device) if stats else None
    rx, rw = slot_for(x, loose[0]), slot_for(w, loose[1])
    _lib.check(lib.embnet_conv2d_stem_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, h, wd, pad, pad, oh, ow, _lib.ptr(st), rx.data_ptr(),
                                          rw.data_ptr(), _lib.stream()))
    return (y, st) if stats else y


def ref64(x, w, pad):
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1)
    return torch.nn.functional.conv2d(xt, wt, stride=2, padding=pad).permute(0, 2, 3, 1).numpy()


def operands(rng, n, h, wd, amplitude=1.0):
    x = np.zeros((n, h, wd, 4), dtype=np.float32)
    x[..., :3] = (rng.standard_normal((n, h, wd, 3)) * amplitude).astype(np.float32)      # (the fourth channel is the zero pad)
    w = np.zeros((7, 7, 4, 64), dtype=np.float32)
    w[:, :, :3, :] = (rng.standard_normal((7, 7, 3, 64)) / 12.0).astype(np.float32)
    return x, w


@pytest.mark.parametrize("n,h,wd,pad", [(2, 224, 224, 3), (3, 64, 64, 3), (2, 75, 61, 3), (1, 23, 39, 0), (5, 128, 128, 3), (40, 96, 96, 3)])
def test_forward_vs_float64(dev, n, h, wd, pad):
    rng = np.random.default_rng(n + h + wd)
    x, w = operands(rng, n, h, wd)
    y = stem(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), pad)
    want = ref64(x, w, pad)
    assert y.shape == want.shape
    err = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    assert np.isfinite(y.cpu().numpy()).all() and err < 1.5e-6, err


@pytest.mark.parametrize("amplitude", [1e-3, 1.0, 1e3])
def test_amplitudes_with_loose_ranges(dev, amplitude):
    rng = np.random.default_rng(9)
    x, w = operands(rng, 3, 80, 80, amplitude)
    y = stem(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), 3, loose=(8.0, 3.0))
    want = ref64(x, w, 3)
    err = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 1.5e-6, err


def test_the_pad_channel_and_the_zero_tap_do_not_leak(dev):
    """A non-zero fourth input channel is multiplied by whatever the kernel holds there (here: something), and huge values right of
    every window (the column the zero tap reads) must not reach the result."""
    rng = np.random.default_rng(2)
    x, w = operands(rng, 2, 48, 48)
    x[..., 3] = 7.0
    w[:, :, 3, :] = 0.25
    y = stem(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), 3)
    want = ref64(x, w, 3)
    assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < 1.5e-6
    x, w = operands(rng, 1, 40, 40)
    x[:, :, 33:, :3] = 6e4                         # finite but enormous: pixels only the zero tap of the last output column's window sees
    y = stem(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), 0)
    want = ref64(x, w, 0)
    assert np.isfinite(y.cpu().numpy()).all()
    assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < 1.5e-6


def test_integer_operands_are_exact(dev):
    torch.manual_seed(4)
    for (n, h, wd, pad) in ((2, 70, 70, 3), (3, 37, 52, 3), (1, 64, 64, 0)):
        x = torch.zeros((n, h, wd, 4), device=dev)
        x[..., :3] = torch.randint(-3, 4, (n, h, wd, 3), device=dev).float()
        w = torch.zeros((7, 7, 4, 64), device=dev)
        w[:, :, :3, :] = torch.randint(-2, 3, (7, 7, 3, 64), device=dev).float()
        y = stem(x, w, pad)
        want = ref64(x.cpu().numpy(), w.cpu().numpy(), pad)
        assert np.array_equal(y.cpu().numpy().astype(np.float64), want)


@pytest.mark.parametrize("n,h,wd", [(4, 224, 224), (3, 75, 61)])
def test_statistics_partials(dev, n, h, wd):
    rng = np.random.default_rng(h)
    x, w = operands(rng, n, h, wd)
    y, st = stem(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), 3, stats=True)
    yv = y.cpu().numpy().astype(np.float64).reshape(-1, 64)
    s = st.cpu().numpy().astype(np.float64)
    assert np.isfinite(s).all()
    assert np.abs(s[0].sum(1) - yv.sum(0)).max() <= 1e-4 * np.abs(yv).sum(0).max()
    assert np.abs(s[1].sum(1) - (yv ** 2).sum(0)).max() <= 1e-5 * (yv ** 2).sum(0).max()
    assert (s[1].max(1) >= (yv ** 2).max(0) * (1 - 1e-6)).all()          # every square is a term of one band's sum


def test_bench_size_vs_gather_kernel_and_repeatable(dev):
    lib = _lib.lib()
    n, h = 128, 224
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.zeros((n, h, h, 4), device=dev)
    x[..., :3] = torch.randn((n, h, h, 3), device=dev, generator=g)
    w = torch.zeros((7, 7, 4, 64), device=dev)
    w[:, :, :3, :] = torch.randn((7, 7, 3, 64), device=dev, generator=g) / 12
    y1 = stem(x, w, 3)
    y2 = stem(x, w, 3)
    assert torch.equal(y1, y2)
    y6 = torch.empty((n, 112, 112, 64), device=dev)
    ws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, 4, 7, 7, 64, 112, 112) // 4, 4), device=dev)
    rx, rw = slot_for(x), slot_for(w)
    _lib.check(lib.embnet_conv2d_fwd_f32_ex(x.data_ptr(), w.data_ptr(), None, y6.data_ptr(), n, h, h, 4, 7, 7, 64, 2, 3, 3, 112, 112, 0, None, None,
                                            None, 0, None, ws.data_ptr(), ws.numel() * 4, rx.data_ptr(), rw.data_ptr(), _lib.stream()))
    err = float((y1 - y6).abs().max() / y6.abs().max())
    assert err < 2e-6, err


def test_both_ranges_are_required_and_other_geometries_refused(dev):
    lib = _lib.lib()
    x = torch.rand((1, 32, 32, 4), device=dev)
    w = torch.rand((7, 7, 4, 64), device=dev)
    y = torch.empty((1, 16, 16, 64), device=dev)
    assert lib.embnet_conv2d_stem_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 32, 32, 3, 3, 16, 16, None, None, None, _lib.stream()) != 0
    assert b"range" in lib.embnet_last_error()
    assert lib.embnet_conv2d_stem_supported(1, 32, 32, 3, 7, 7, 64, 2, 3, 3, 16, 16) == 0      # three channels: the caller pads to four
    assert lib.embnet_conv2d_stem_supported(1, 32, 32, 4, 7, 7, 32, 2, 3, 3, 16, 16) == 0
    assert lib.embnet_conv2d_stem_supported(1, 32, 32, 4, 7, 7, 64, 1, 3, 3, 32, 32) == 0


def test_resnet18_runs_its_stem_on_the_kernel(dev):
    from embeddingnet_amd.backbones import get_backbone

    def run(on):
        old = L.STEM_CONV[0]
        L.STEM_CONV[0] = on
        try:
            torch.manual_seed(0)
            base, _ = get_backbone((96, 96, 3), encodings_len=32, backbone_name="resnet18", backbone_weights=None, seed=2, device=dev)
            base.train()
            g = torch.Generator().manual_seed(3)
            x = torch.rand((8, 96, 96, 3), generator=g).to(dev)
            t = torch.randn((8, 32), generator=g).to(dev)
            _lib.trace_reset(); _lib.trace_enable(True)
            try:
                y = base(x)
                (y * t).sum().backward()
                names = [r[0] for r in _lib.trace_records()]
            finally:
                _lib.trace_enable(False)
            assert not L.current_context().leftovers()
            return y.detach(), [p.grad.clone() for p in base.parameters() if p.grad is not None], names
        finally:
            L.STEM_CONV[0] = old

    y1, g1, n1 = run(True)
    y0, g0, n0 = run(False)
    assert sum("conv_stem_kernel" in s for s in n1) == 1 and not any("conv_stem_kernel" in s for s in n0)
    assert sum("conv_fwd_h_kernel" in s for s in n0) == sum("conv_fwd_h_kernel" in s for s in n1) + 1
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(y1, y0) < 2e-5
    gmax = max(float(b.abs().max()) for b in g0)
    for a, b in zip(g1, g0):
        # (gradients that are analytically zero — a conv bias-like parameter in front of a BatchNormalization — are rounding noise in
        # both arithmetics: measured against the model's largest gradient)
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-2 * gmax), (rel(a, b), float(b.abs().max()), gmax)
