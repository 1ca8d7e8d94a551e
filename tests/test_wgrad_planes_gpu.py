"""Parity of the planes-based weight gradient (csrc/conv_wgrad_planes.hip) on the GPU, through the C ABI:

  * against a float64 weight gradient of the SAME operands (the planes decode exactly to the fp32 tensors), on every
    ResNet18/34/50 3x3 stride-1 geometry class, on ragged maps (odd sizes, one image, a map smaller than a stage) and with
    the position range split over many work-groups and over one;
  * against the gather-loop weight gradient of conv.hip (embnet_conv2d_wgrad_f32) at the bench's layer sizes;
  * bitwise reproducible from launch to launch (fixed-order slab sum, no atomics).

Reference behaviour being matched: the kernel gradient of keras Conv2D(3x3, padding 1 after ZeroPadding2D) inside
image-classifiers' ResNet (reference embedding_net/backbones.py:99-104).
"""
import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    return torch.device("cuda", 0)


def planes_of(x):
    c = x.shape[-1]
    p = torch.empty(3 * x.numel(), device=x.device, dtype=torch.int16)
    _lib.check(_lib.lib().embnet_planes_from_f32(x.data_ptr(), x.numel() // c, c, p.data_ptr(), _lib.stream()))
    return p


def wgrad_planes(x, dy, reduce=1):
    lib = _lib.lib()
    n, h, w, c = x.shape
    k = dy.shape[-1]
    assert lib.embnet_conv2d_wgrad_planes_supported(n, h, w, c, 3, 3, k, 1, 1, 1, h, w) == 1
    ws = torch.empty(max(lib.embnet_conv2d_wgrad_planes_workspace_bytes(n, h, w, c, k) // 4, 4), device=x.device)
    dw = torch.full((3, 3, c, k), float("nan"), device=x.device)
    xp, dp = planes_of(x), planes_of(dy)                   # (held: a temporary's block would be handed to the next allocation)
    _lib.check(lib.embnet_conv2d_wgrad_planes_f32(xp.data_ptr(), dp.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                                  ws.numel() * 4, n, h, w, c, k, reduce, _lib.stream()))
    return dw, ws


def wgrad_gather(x, dy):
    lib = _lib.lib()
    n, h, w, c = x.shape
    k = dy.shape[-1]
    ws = torch.empty(max(lib.embnet_conv2d_wgrad_workspace_bytes(n, c, 3, 3, k, h, w) // 4, 4), device=x.device)
    dw = torch.empty((3, 3, c, k), device=x.device)
    _lib.check(lib.embnet_conv2d_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, n, h, w, c,
                                           3, 3, k, 1, 1, 1, h, w, None, None, 0, _lib.stream()))
    return dw


def wgrad64(x, dy):
    """float64 dW[r,s,c,k] = sum_{n,oh,ow} x[n, oh+r-1, ow+s-1, c] dy[n, oh, ow, k] (zero padding)."""
    xt = torch.from_numpy(x.astype(np.float64)).permute(3, 0, 1, 2)          # [C, N, H, W]: channels as the batch
    dt = torch.from_numpy(dy.astype(np.float64)).permute(3, 0, 1, 2)         # [K, N, H, W]: filters
    g = torch.nn.functional.conv2d(xt, dt, padding=1)                         # [C, K, 3, 3]
    return g.permute(2, 3, 0, 1).numpy()


GEOMS = [  # n, h, w, c, k
    (4, 56, 56, 64, 64),       # ResNet18/34 stage 0 (halo of 4 blocks: the ring is exactly full)
    (4, 28, 28, 128, 128),
    (6, 14, 14, 256, 256),
    (8, 7, 7, 512, 512),       # few positions, 64 tiles: few stages per split
    (2, 56, 56, 64, 128),
    (3, 13, 9, 64, 192),       # ragged: positions not a multiple of the stage
    (1, 5, 5, 64, 64),         # one image smaller than two stages: a single split
    (1, 1, 1, 64, 64),         # one pixel: only the centre tap sees it
    (2, 20, 31, 128, 64),
    (5, 3, 62, 64, 64),        # widest supported map (window of 4 blocks + run-ahead = the whole ring)
]


@pytest.mark.parametrize("n,h,w,c,k", GEOMS)
def test_wgrad_planes_vs_float64(dev, n, h, w, c, k):
    torch.manual_seed(n * 1000 + h)
    x = torch.randn(n, h, w, c, device=dev) * torch.logspace(-2, 2, c, device=dev)
    dy = torch.randn(n, h, w, k, device=dev) * torch.logspace(1, -1, k, device=dev)
    dw, _ = wgrad_planes(x, dy)
    ref = wgrad64(x.cpu().numpy(), dy.cpu().numpy())
    # scale of a sum of n*h*w products, per (c, k): |x|.|dy| summed; fp32 accumulation over the positions
    mag = np.einsum("nhwc,nhwk->ck", np.abs(x.cpu().numpy()).astype(np.float64), np.abs(dy.cpu().numpy()).astype(np.float64))
    err = np.abs(dw.cpu().numpy() - ref) / (mag[None, None] + 1e-30)
    assert np.isfinite(dw.cpu().numpy()).all()
    assert err.max() < 2e-6, err.max()                     # fp32 products (six-term split), fp32 sums: ~1e-7 typical


def test_wgrad_planes_padding_taps_are_exact(dev):
    """Integer operands: every product and sum is exact, so any position / tap / padding slip shows as an integer error."""
    torch.manual_seed(3)
    n, h, w, c, k = 3, 9, 11, 64, 64
    x = torch.randint(-3, 4, (n, h, w, c), device=dev).float()
    dy = torch.randint(-3, 4, (n, h, w, k), device=dev).float()
    dw, _ = wgrad_planes(x, dy)
    assert np.array_equal(dw.cpu().numpy().astype(np.float64), wgrad64(x.cpu().numpy(), dy.cpu().numpy()))


@pytest.mark.parametrize("n,h,w,c,k", [(128, 56, 56, 64, 64), (128, 28, 28, 128, 128), (128, 14, 14, 256, 256), (128, 7, 7, 512, 512)])
def test_wgrad_planes_vs_gather_kernel_at_bench_sizes(dev, n, h, w, c, k):
    torch.manual_seed(h)
    x = torch.relu(torch.randn(n, h, w, c, device=dev))
    dy = torch.randn(n, h, w, k, device=dev) * 1e-3
    a, _ = wgrad_planes(x, dy)
    b = wgrad_gather(x, dy)
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() < 2e-5 * scale       # two fp32 summation orders over 6 272 ... 401 408 pixels
    a2, _ = wgrad_planes(x, dy)
    assert torch.equal(a, a2)                              # fixed-order slab sum: bitwise reproducible


def test_wgrad_planes_slabs_for_the_deferred_sum(dev):
    """reduce = 0 leaves [splits][9ck] slabs whose fixed-order sum (embnet_slab_reduce_multi) is the gradient."""
    lib = _lib.lib()
    torch.manual_seed(5)
    n, h, w, c, k = 8, 14, 14, 128, 64
    x, dy = torch.randn(n, h, w, c, device=dev), torch.randn(n, h, w, k, device=dev)
    full, _ = wgrad_planes(x, dy, reduce=1)
    raw, ws = wgrad_planes(x, dy, reduce=0)
    splits = lib.embnet_conv2d_wgrad_planes_splits(n, h, w, c, k)
    assert splits > 1 and torch.isnan(raw).all()           # dw untouched until the sum runs
    rows = np.asarray([(ws.data_ptr(), raw.data_ptr(), 9 * c * k, splits)], dtype=np.int64)
    _lib.check(lib.embnet_slab_reduce_multi(rows.ctypes.data, 1, _lib.stream()))
    assert torch.equal(raw, full)


def test_wgrad_planes_unsupported_geometries_are_refused(dev):
    lib = _lib.lib()
    assert lib.embnet_conv2d_wgrad_planes_supported(8, 56, 56, 64, 3, 3, 64, 2, 1, 1, 28, 28) == 0      # stride 2
    assert lib.embnet_conv2d_wgrad_planes_supported(8, 56, 56, 64, 1, 1, 64, 1, 0, 0, 56, 56) == 0      # 1x1
    assert lib.embnet_conv2d_wgrad_planes_supported(8, 56, 56, 48, 3, 3, 64, 1, 1, 1, 56, 56) == 0      # c % 64
    assert lib.embnet_conv2d_wgrad_planes_supported(8, 112, 112, 64, 3, 3, 64, 1, 1, 1, 112, 112) == 0  # window exceeds the ring
    x = torch.zeros(4, device=dev)
    rc = lib.embnet_conv2d_wgrad_planes_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 0, 8, 112, 112, 64, 64, 1, _lib.stream())
    assert rc != 0 and b"unsupported" in lib.embnet_last_error()
