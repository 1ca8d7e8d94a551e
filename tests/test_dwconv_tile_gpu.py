"""Stride-1 depthwise convolution on small maps from an LDS tile (csrc/dwconv_tile.hip: EfficientNet's 28x28 ... 7x7 MBConv stages,
reference embedding_net/backbones.py:84-98 via efficientnet's MBConv) through the C ABI against float64 arithmetic:
forward (with and without the BatchNorm statistics partials), the data gradient (with and without the BatchNorm-backward sums of
the layer in front), on whole-image units (one image, groups of images with a ragged last group, two passes per tile), on row
bands, with four- and eight-quad channel chunks, and the kernel trace showing that the tile kernel is what ran."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from embeddingnet_amd import _lib
from embeddingnet_amd._lib import check, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _names(fn):
    _lib.trace_reset(); _lib.trace_enable(True)
    out = fn()
    torch.cuda.synchronize()
    names = [r[0] for r in _lib.trace_records()]
    _lib.trace_enable(False)
    return out, names


def _ref_fwd(x, w, k):
    """float64 'same' depthwise correlation, NHWC (Keras DepthwiseConv2D(padding='same', strides=1))."""
    c = x.shape[-1]
    xt = x.double().permute(0, 3, 1, 2)
    wt = w.double().reshape(k, k, c).permute(2, 0, 1).unsqueeze(1)
    return F.conv2d(xt, wt, padding=(k - 1) // 2, groups=c).permute(0, 2, 3, 1).contiguous()


def _err(got, want):
    return (got.double().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-30)


# (n, h, w, c, k): 14x14 one image per unit (8 quads); 7x7 groups of four images, ragged last group; 28x28 four-quad chunks, two
# passes; 14x14 with C % 32 != 0 (four quads, four images per unit); a map that is not a multiple of 7 wide; 30x30: row bands;
# 2 images only; many images (several units per workgroup)
SHAPES = [(5, 14, 14, 64, 3), (5, 14, 14, 64, 5), (6, 7, 7, 96, 5), (7, 7, 7, 1152, 3), (3, 28, 28, 48, 5), (3, 28, 28, 144, 3),
          (9, 14, 14, 240, 5), (4, 12, 10, 32, 3), (2, 30, 30, 48, 5), (3, 30, 30, 16, 3), (2, 7, 7, 32, 5), (70, 14, 14, 32, 5),
          (130, 7, 7, 64, 3)]


@pytest.mark.parametrize("n,h,w,c,k", SHAPES)
def test_forward_and_statistics_vs_float64(dev, n, h, w, c, k):
    lib = _lib.lib()
    torch.manual_seed(n * 1000 + h * 10 + k)
    x = torch.randn(n, h, w, c, device=dev)
    wt = torch.randn(k, k, c, 1, device=dev) * 0.3
    pad = (k - 1) // 2
    want = _ref_fwd(x.cpu(), wt.cpu(), k)
    y = torch.full((n, h, w, c), float("nan"), device=dev)
    _, names = _names(lambda: check(lib.embnet_dwconv2d_fwd_f32(x.data_ptr(), wt.data_ptr(), y.data_ptr(), n, h, w, c, k, k, 1, pad, pad, h, w, stream())))
    assert names == ["embnet::dwt::dw_tile_kernel"], names
    assert _err(y, want) < 2e-6
    rows = lib.embnet_dwconv2d_fwd_stats_rows(n, c, k, k, 1, h, w)
    assert rows > 0
    stats = torch.full((2, c, rows), float("nan"), device=dev)
    y2 = torch.empty_like(y)
    check(lib.embnet_dwconv2d_fwd_stats_f32(x.data_ptr(), wt.data_ptr(), y2.data_ptr(), n, h, w, c, k, k, 1, pad, pad, h, w, stats.data_ptr(), stream()))
    assert torch.equal(y2, y)
    s = stats.double().sum(-1).cpu()
    assert torch.isfinite(stats).all()
    want1, want2 = want.sum((0, 1, 2)), (want ** 2).sum((0, 1, 2))
    assert (s[0] - want1).abs().max().item() <= 1e-5 * want.abs().sum((0, 1, 2)).max().item()
    assert (s[1] - want2).abs().max().item() <= 1e-5 * want2.max().item()


@pytest.mark.parametrize("n,h,w,c,k", SHAPES)
@pytest.mark.parametrize("act", [2, 1])
def test_data_gradient_and_batchnorm_backward_sums_vs_float64(dev, n, h, w, c, k, act):
    """dx = correlate(dy, flipped w); with the sums: dz = dx * act'(scale * e + shift), partial rows of sum dz and
    sum dz * (e - mean) * rstd (what embnet_bn_bwd's reduction pass would produce for the BatchNormalization in front)."""
    lib = _lib.lib()
    torch.manual_seed(n * 1000 + h * 10 + k + act)
    dy = torch.randn(n, h, w, c, device=dev)
    wt = torch.randn(k, k, c, 1, device=dev) * 0.3
    e = torch.randn(n, h, w, c, device=dev)
    scale = torch.linspace(0.5, 1.5, c, device=dev); shift = torch.linspace(-0.3, 0.3, c, device=dev)
    mean = torch.linspace(-0.1, 0.1, c, device=dev); rstd = torch.linspace(0.8, 1.2, c, device=dev)
    pad = (k - 1) // 2
    # float64: dx through autograd of the forward correlation
    xr = torch.zeros(n, h, w, c, dtype=torch.float64, requires_grad=True)
    _ref_fwd(xr, wt.cpu(), k).backward(dy.cpu().double())
    want = xr.grad
    dx = torch.full((n, h, w, c), float("nan"), device=dev)
    _, names = _names(lambda: check(lib.embnet_dwconv2d_dgrad_f32(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), n, h, w, c, k, k, 1, pad, pad, h, w, stream())))
    assert names == ["embnet::dwt::dw_tile_kernel"], names
    assert _err(dx, want) < 2e-6
    rows = lib.embnet_dwconv2d_dgrad_bnsums_rows(n, h, w, c, k, k, 1)
    assert rows > 0
    part = torch.full((2, c, rows), float("nan"), device=dev)
    dx2 = torch.empty_like(dx)
    check(lib.embnet_dwconv2d_dgrad_bnsums_f32(dy.data_ptr(), wt.data_ptr(), dx2.data_ptr(), n, h, w, c, k, k, 1, pad, pad, h, w, e.data_ptr(),
                                               scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), act, part.data_ptr(), rows, stream()))
    assert torch.equal(dx2, dx) and torch.isfinite(part).all()
    z = e.cpu().double() * scale.cpu().double() + shift.cpu().double()
    if act == 1:
        dz = want * (z > 0)
    else:
        sg = torch.sigmoid(z)
        dz = want * (sg + z * sg * (1 - sg))
    ehat = (e.cpu().double() - mean.cpu().double()) * rstd.cpu().double()
    s = part.double().sum(-1).cpu()
    assert (s[0] - dz.sum((0, 1, 2))).abs().max().item() <= 1e-5 * dz.abs().sum((0, 1, 2)).max().item()
    assert (s[1] - (dz * ehat).sum((0, 1, 2))).abs().max().item() <= 1e-5 * (dz * ehat).abs().sum((0, 1, 2)).max().item()


@pytest.mark.parametrize("n,h,w,c,k", SHAPES)
def test_weight_gradient_vs_float64(dev, n, h, w, c, k):
    """dW[r,s,c] = sum x[o + (r,s) - pad] dy[o]: the tile kernel's slabs + the slab sum, against float64 autograd; tiles that take
    two passes (the 30x30 bands) stay on the wave kernel."""
    lib = _lib.lib()
    torch.manual_seed(n * 1000 + h * 10 + k + 7)
    x = torch.randn(n, h, w, c, device=dev)
    dy = torch.randn(n, h, w, c, device=dev)
    pad = (k - 1) // 2
    wr = torch.zeros(k, k, c, 1, dtype=torch.float64, requires_grad=True)
    _ref_fwd(x.cpu(), wr, k).backward(dy.cpu().double())
    want = wr.grad
    ws = torch.empty(max(lib.embnet_dwconv2d_wgrad_workspace_bytes(n, c, k, k, h, w) // 4, 4), device=dev)
    dw = torch.full((k, k, c, 1), float("nan"), device=dev)
    _, names = _names(lambda: check(lib.embnet_dwconv2d_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                                  n, h, w, c, k, k, 1, pad, pad, h, w, stream())))
    two_pass = h == 30                       # 15-row bands x 5 column blocks = 75 units for 64 slots
    assert names[-1] == "embnet::dw_slab_sum_kernel" and ("dw_tile_wgrad" in names[0]) == (not two_pass), names
    assert _err(dw, want) < 5e-6


def test_asymmetric_padding_and_what_stays_on_the_row_kernels(dev):
    """pad_t / pad_l other than (k - 1) / 2 (the C ABI takes them; same-size output) run on the tile kernel too; stride 2, maps
    above 32 x 32 and C % 16 != 0 stay on the per-thread row kernels."""
    lib = _lib.lib()
    n, h, w, c, k = 3, 14, 14, 32, 5
    x = torch.randn(n, h, w, c, device=dev)
    wt = torch.randn(k, k, c, 1, device=dev)
    y = torch.empty_like(x)
    _, names = _names(lambda: check(lib.embnet_dwconv2d_fwd_f32(x.data_ptr(), wt.data_ptr(), y.data_ptr(), n, h, w, c, k, k, 1, 1, 3, h, w, stream())))
    assert names == ["embnet::dwt::dw_tile_kernel"]
    xt = F.pad(x.cpu().double().permute(0, 3, 1, 2), (3, 1, 1, 3))
    want = F.conv2d(xt, wt.cpu().double().reshape(k, k, c).permute(2, 0, 1).unsqueeze(1), groups=c).permute(0, 2, 3, 1)
    assert _err(y, want) < 2e-6
    for (hh, cc, st) in [(14, 32, 2), (56, 32, 1), (14, 24, 1)]:
        x = torch.randn(2, hh, hh, cc, device=dev)
        wt = torch.randn(3, 3, cc, 1, device=dev)
        oh = (hh + st - 1) // st
        y = torch.empty(2, oh, oh, cc, device=dev)
        pt = max((oh - 1) * st + 3 - hh, 0) // 2
        _, names = _names(lambda: check(lib.embnet_dwconv2d_fwd_f32(x.data_ptr(), wt.data_ptr(), y.data_ptr(), 2, hh, hh, cc, 3, 3, st, pt, pt, oh, oh, stream())))
        assert names and "dw_tile" not in names[0], (hh, cc, st, names)


_FULL_SIZE_SCRIPT = r"""
import json, sys, torch
from embeddingnet_amd import _lib
from embeddingnet_amd._lib import check, stream
lib = _lib.lib(); dev = torch.device("cuda:0"); out = {}
for (h, c, k) in [(28, 240, 5), (14, 480, 3), (14, 672, 5), (7, 1152, 5)]:
    n, pad = 256, (k - 1) // 2
    g = torch.Generator(device=dev).manual_seed(h * 1000 + c)
    x = torch.randn(n, h, h, c, device=dev, generator=g); dy = torch.randn(n, h, h, c, device=dev, generator=g)
    w = torch.randn(k, k, c, 1, device=dev, generator=g) * 0.2
    y = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
    rows = lib.embnet_dwconv2d_fwd_stats_rows(n, c, k, k, 1, h, h)
    st = torch.zeros(2, c, rows, device=dev)
    check(lib.embnet_dwconv2d_fwd_stats_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, h, h, c, k, k, 1, pad, pad, h, h, st.data_ptr(), stream()))
    check(lib.embnet_dwconv2d_dgrad_f32(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, h, h, c, k, k, 1, pad, pad, h, h, stream()))
    ws = torch.empty(max(lib.embnet_dwconv2d_wgrad_workspace_bytes(n, c, k, k, h, h) // 4, 4), device=dev)
    check(lib.embnet_dwconv2d_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, n, h, h, c, k, k, 1, pad, pad, h, h, stream()))
    torch.cuda.synchronize()
    pick = lambda t: t.flatten()[:: max(t.numel() // 4096, 1)][:4096].double().cpu().tolist()
    out[f"{h}x{c}k{k}"] = dict(y=pick(y), dx=pick(dx), dw=dw.flatten().double().cpu().tolist(), s=st.double().sum(-1).flatten().cpu().tolist(),
                               ysum=float(y.double().sum()), dxsum=float(dx.double().sum()))
json.dump(out, open(sys.argv[1], "w"))
"""


def test_tile_kernels_equal_row_kernels_at_full_size(dev, tmp_path):
    """BASELINE config C5's stride-1 small-map layers at the full local batch (256 images): forward + statistics, data gradient and
    weight gradient from the LDS-tile kernels against the per-thread row kernels (EMBNET_DW_TILE=0; the knob is read once per
    process, so each side runs in a child process on the same seeded tensors): sampled elements, whole-tensor sums, every
    statistic and every weight-gradient element within fp32 summation order."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for on in ("1", "0"):
        path = tmp_path / f"dw_{on}.json"
        env = dict(os.environ, EMBNET_DW_TILE=on, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.run([sys.executable, "-c", _FULL_SIZE_SCRIPT, str(path)], check=True, env=env, cwd=root, timeout=600)
        res[on] = json.load(open(path))
    for layer, a in res["1"].items():
        b = res["0"][layer]
        for key in ("y", "dx", "dw", "s"):
            ta, tb = torch.tensor(a[key]), torch.tensor(b[key])
            assert float((ta - tb).abs().max() / tb.abs().max().clamp_min(1e-30)) < 2e-5, (layer, key)
        for key in ("ysum", "dxsum"):
            assert abs(a[key] - b[key]) <= 1e-6 * max(abs(b[key]), 1.0) + 1e-2, (layer, key, a[key], b[key])
