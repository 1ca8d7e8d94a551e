"""Parity of the patch convolution (csrc/conv_patch.hip: pre-split bf16 planes, LDS patch + weight ring) on the GPU,
through the C ABI:

  * the planes a BatchNormalization writes decode EXACTLY to its fp32 output (the three-way truncation split loses nothing);
  * forward and stride-1 data gradient against a float64 convolution of the same operands, on every ResNet18/34 3x3
    geometry class (full tiles, ragged tails, K split across work-groups) and with each epilogue (bias, ReLU, residual,
    per-channel sums);
  * the layer path (BatchNormalization(planes_for=conv) -> Conv2D) against the same modules with the patch path switched
    off (gather kernels), values and all gradients;
  * a ResNet18 training step with the patch convs against the step with the gather convs.

Reference behaviour being matched: keras Conv2D / BatchNormalization inside image-classifiers' ResNet
(reference embedding_net/backbones.py:99-104).
"""
import os

import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib
from embeddingnet_amd import backbones as B
from embeddingnet_amd import layers as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    return torch.device("cuda", 0)


def planes_of(x):
    """bf16 planes [3][C/16][pixels][16] of an NHWC fp32 tensor (embnet_planes_from_f32)."""
    c = x.shape[-1]
    p = torch.empty(3 * x.numel(), device=x.device, dtype=torch.int16)
    _lib.check(_lib.lib().embnet_planes_from_f32(x.data_ptr(), x.numel() // c, c, p.data_ptr(), _lib.stream()))
    return p


F16 = os.environ.get("EMBNET_PLANES_F16", "1") != "0"      # the planes' format of this process (csrc/gemm_engine.h, csrc/common.h)


def pieces(raw16, lead):
    """uint16 planes [3][...] -> float64 sum of the pieces: three bf16 pieces, or (EMBNET_PLANES_F16=1) two fp16 pieces / s with
    (s, 1 / s) in the first two floats of the third plane's space."""
    flat = raw16.reshape(3, -1)
    if F16:
        s, inv = flat[2][:4].view(np.float32)[:2]
        assert s > 0 and s * inv == 1.0 and np.log2(s) == np.round(np.log2(s)), (s, inv)
        return (flat[0].view(np.float16).astype(np.float64) + flat[1].view(np.float16).astype(np.float64)).reshape(lead) * float(inv)
    return (flat.astype(np.uint32) << 16).view(np.float32).astype(np.float64).sum(axis=0).reshape(lead)


def decode(planes, shape):
    """planes -> float64 NHWC."""
    c = shape[-1]
    m = int(np.prod(shape)) // c
    f = pieces(planes.cpu().numpy().view(np.uint16), (c // 16, m, 16))         # [c/16][m][16]
    return f.transpose(1, 0, 2).reshape(shape)


def conv64(x, w, pad):
    """float64 stride-1 cross-correlation, NHWC x RSCK, zero padding `pad` on every side."""
    y = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), padding=pad)
    return y.permute(0, 2, 3, 1).numpy()


def dgrad64(dy, w, pad):
    r = w.shape[0]
    wf = np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))       # flipped, channels swapped
    return conv64(dy, wf, r - 1 - pad)


def test_planes_decode_exactly(dev):
    torch.manual_seed(0)
    x = torch.randn(3, 9, 7, 48, device=dev) * torch.logspace(-6, 6, 48, device=dev)
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, 3.0e38], device=dev)     # (fp32 subnormals lose their low bits: below bf16's range)
    p = planes_of(x)
    torch.cuda.synchronize()
    if F16:
        # two fp16 pieces of x s, the tensor's largest |x| scaled into [2^14, 2^15): 22 bits of every element that is within
        # 2^-14 of the largest, absolute error <= 2^-25 of the scaled range below that (this tensor spans 10^12 by design)
        got, want = decode(p, x.shape), x.cpu().numpy().astype(np.float64)
        top = np.abs(want).max()
        err = np.abs(got - want)
        assert (err <= np.maximum(np.abs(want) * 2.0 ** -21, top * 2.0 ** -39)).all()
        return
    assert np.array_equal(decode(p, x.shape), x.cpu().numpy().astype(np.float64))


GEOMS = [  # n, h, w, c, k: the ResNet 3x3 stride-1 classes + ragged ones
    (8, 56, 56, 64, 64),       # 56^2 x 64 (BN=64 tiles)
    (4, 56, 56, 64, 128),
    (8, 28, 28, 128, 128),
    (16, 14, 14, 256, 256),
    (32, 7, 7, 512, 512),      # few tiles: reduction split across work-groups + fix-up
    (3, 13, 9, 32, 96),        # ragged rows, K not a multiple of the tile
    (1, 5, 5, 16, 32),
    (2, 20, 31, 80, 160),
]


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_patch_forward_and_data_gradient_vs_float64(dev, geom):
    n, h, w, c, k = geom
    lib = _lib.lib()
    if not lib.embnet_conv2d_patch_supported(n, c, 3, 3, k, 1, h, w):
        pytest.skip("geometry not served by the patch kernel")
    rng = np.random.default_rng(sum(geom))
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, c, k)) / np.sqrt(9 * c)).astype(np.float32)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(wt).to(dev)
    y = torch.empty((n, h, w, k), device=dev)
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(n, c, 3, 3, k, h, w), 4) // 4, device=dev)
    xp = planes_of(xd)
    _lib.check(lib.embnet_conv2d_patch_f32(xp.data_ptr(), L.weight_planes(wd, 0).data_ptr(), None, y.data_ptr(), n, h, w, c, 3, 3, k,
                                           1, 1, h, w, 0, None, None, ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    want = conv64(x.astype(np.float64), wt.astype(np.float64), 1)
    err = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 3e-6, err                       # fp32 accumulation over 9*C terms; the split itself is < 2^-20

    if lib.embnet_conv2d_patch_supported(n, k, 3, 3, c, 1, h, w):
        dy = rng.standard_normal((n, h, w, k)).astype(np.float32)
        dyd = torch.from_numpy(dy).to(dev)
        add = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(dev)
        dx = torch.empty((n, h, w, c), device=dev)
        L._patch_dgrad(planes_of(dyd), wd, dx, n, h, w, c, 3, 3, k, 1, 1, h, w, add)
        want = dgrad64(dy.astype(np.float64), wt.astype(np.float64), 1) + add.cpu().numpy()
        err = np.abs(dx.cpu().numpy() - want).max() / np.abs(want).max()
        assert err < 3e-6, err


@pytest.mark.parametrize("n,h,w,c,k", [(8, 28, 28, 128, 128), (5, 11, 13, 64, 96), (32, 7, 7, 512, 512)])
def test_patch_epilogues(dev, n, h, w, c, k):
    """bias + ReLU + residual, and the per-channel sums / sums of squares a following BatchNormalization consumes."""
    lib = _lib.lib()
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, c, k)) / np.sqrt(9 * c)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    res = rng.standard_normal((n, h, w, k)).astype(np.float32)
    xd, wd, bd, rd = (torch.from_numpy(a).to(dev) for a in (x, wt, bias, res))
    rows = lib.embnet_conv2d_patch_stats_rows(n, h, w)
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(n, c, 3, 3, k, h, w), 4) // 4, device=dev)
    base = conv64(x.astype(np.float64), wt.astype(np.float64), 1)
    xp = planes_of(xd)
    for relu, use_bias, use_res in [(1, True, True), (0, False, True), (1, True, False)]:
        y = torch.empty((n, h, w, k), device=dev)
        stats = torch.full((2, k, rows), float("nan"), device=dev)
        _lib.check(lib.embnet_conv2d_patch_f32(xp.data_ptr(), L.weight_planes(wd, 0).data_ptr(), bd.data_ptr() if use_bias else None,
                                               y.data_ptr(), n, h, w, c, 3, 3, k, 1, 1, h, w, relu, rd.data_ptr() if use_res else None,
                                               stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, _lib.stream()))
        want = base + (bias if use_bias else 0)
        if relu:
            want = np.maximum(want, 0)                  # the conv's own activation; the Add layer comes after it
        if use_res:
            want = want + res
        got = y.cpu().numpy()
        assert np.abs(got - want).max() / np.abs(want).max() < 3e-6
        st = stats.cpu().numpy().astype(np.float64).sum(axis=2)             # [2][K]
        flat = got.reshape(-1, k).astype(np.float64)
        assert np.allclose(st[0], flat.sum(0), rtol=1e-5, atol=1e-3 * np.sqrt(flat.shape[0]))
        assert np.allclose(st[1], (flat ** 2).sum(0), rtol=1e-5)


@pytest.mark.parametrize("n,h,c,k", [(128, 56, 64, 64), (128, 28, 128, 128), (128, 14, 256, 256), (128, 7, 512, 512), (96, 28, 128, 256)])
def test_patch_at_bench_sizes_vs_gather_kernel_and_repeatable(dev, n, h, c, k):
    """BASELINE config C2's layers (local batch 128): whole rounds of tiles plus a remainder cut along the channel chunks
    (partial tiles + fix-up).  Against the gather kernel (itself checked against float64 at small sizes), with the following
    BatchNormalization's sums, and bit-identical from launch to launch."""
    lib = _lib.lib()
    gen = torch.Generator().manual_seed(n + h)
    x = torch.randn(n, h, h, c, generator=gen).to(dev)
    conv = L.Conv2D(c, k, 3, padding=1, use_bias=False, kernel_initializer="he_uniform", gen=gen).to(dev)
    res = torch.randn(n, h, h, k, generator=gen).to(dev)
    with torch.no_grad():
        L.PATCH_CONV[0] = False
        try:
            want = conv(x, residual=res, emit_stats=True)
        finally:
            L.PATCH_CONV[0] = True
        x._planes = planes_of(x)
        outs = [conv(x, residual=res, emit_stats=True) for _ in range(3)]
    torch.cuda.synchronize()
    assert getattr(outs[0], "_wants_dy_planes", False)
    assert float((outs[0] - want).abs().max() / want.abs().max()) < 5e-6      # two fp32 accumulation orders over up to 4608 terms
    for o in outs[1:]:
        assert torch.equal(o, outs[0]) and torch.equal(o._bn_partials, outs[0]._bn_partials)
    m = n * h * h
    for j, ref in enumerate((want.reshape(m, k).double().sum(0), want.reshape(m, k).double().pow(2).sum(0))):
        got = outs[0]._bn_partials[j].double().sum(-1)
        assert float((got - ref).abs().max() / ref.abs().max().clamp_min(1.0)) < 1e-5


class _Pair(torch.nn.Module):
    """BatchNormalization(+ReLU) -> 3x3 conv -> BatchNormalization -> 3x3 conv (+ identity shortcut): the pattern of a
    basic ResNet unit, exercising planes emission in forward AND of the BN input gradient in backward."""

    def __init__(self, c, k, gen):
        super().__init__()
        self.bn1 = L.BatchNormalization(c, epsilon=2e-5, relu=True)
        self.conv1 = L.Conv2D(c, k, 3, padding=1, use_bias=False, kernel_initializer="he_uniform", gen=gen)
        self.bn2 = L.BatchNormalization(k, epsilon=2e-5, relu=True)
        self.conv2 = L.Conv2D(k, c, 3, padding=1, use_bias=False, kernel_initializer="he_uniform", gen=gen)

    def forward(self, x):
        a, sc = self.bn1(x, with_skip=True, planes_for=self.conv1)
        y = self.bn2(self.conv1(a, emit_stats=True), planes_for=self.conv2)
        return self.conv2(y, residual=sc, emit_stats=True)


@pytest.mark.parametrize("n,h,c,k", [(8, 28, 128, 128), (6, 14, 64, 96)])
def test_unit_with_planes_equals_unit_on_gather_kernels(dev, n, h, c, k):
    gen = torch.Generator().manual_seed(3)
    net = _Pair(c, k, gen).to(dev).train()
    x0 = torch.randn(n, h, h, c, generator=gen).to(dev)
    dy = torch.randn(n, h, h, c, generator=gen).to(dev)
    out = {}
    for on in (True, False):
        L.PATCH_CONV[0] = on
        try:
            x = x0.clone().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            y = net(x)
            assert bool(getattr(y, "_wants_dy_planes", False)) == on
            y.backward(dy)
            torch.cuda.synchronize()
            out[on] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        finally:
            L.PATCH_CONV[0] = True
    assert not L.DY_PLANES                         # every planes tensor left for a consumer was collected
    for a, b in zip(out[True], out[False]):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        assert rel < 2e-5, rel                     # same operands and split; fp32 accumulation order differs

@pytest.mark.parametrize("n,h,w,c,k", [(8, 28, 28, 128, 128), (5, 11, 13, 64, 96), (32, 7, 7, 512, 512), (128, 14, 14, 256, 256)])
@pytest.mark.parametrize("act", [1, 0])
def test_patch_data_gradient_emits_batchnorm_backward_sums(dev, n, h, w, c, k, act):
    """embnet_conv2d_patch_bnsums_f32: the data gradient of a 3x3 conv whose input was act(scale * e + shift), and the
    BatchNorm-backward sums  sum dz,  sum dz * (e - mean) * rstd  (dz = dx * act'(...)) from the kernel's epilogue (whole tiles and
    the fix-up pass over split tiles), against float64; dx bit for bit what the plain data gradient writes."""
    lib = _lib.lib()
    rng = np.random.default_rng(n + h + act)
    wt = (rng.standard_normal((3, 3, c, k)) / np.sqrt(9 * c)).astype(np.float32)
    dy = rng.standard_normal((n, h, w, k)).astype(np.float32)
    e = rng.standard_normal((n, h, w, c)).astype(np.float32)
    scale, shift = np.linspace(0.5, 1.5, c).astype(np.float32), np.linspace(-0.3, 0.3, c).astype(np.float32)
    mean, rstd = np.linspace(-0.1, 0.1, c).astype(np.float32), np.linspace(0.8, 1.2, c).astype(np.float32)
    wd, dyd, ed = (torch.from_numpy(a).to(dev) for a in (wt, dy, e))
    vec = [torch.from_numpy(a).to(dev) for a in (scale, shift, mean, rstd)]
    dyp = planes_of(dyd)
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(n, k, 3, 3, c, h, w), 4) // 4, device=dev)
    plain = torch.empty((n, h, w, c), device=dev)
    L._patch_dgrad(dyp, wd, plain, n, h, w, c, 3, 3, k, 1, 1, h, w, None)
    rows = lib.embnet_conv2d_patch_stats_rows(n, h, w)
    part = torch.full((2, c, rows), float("nan"), device=dev)
    dx = torch.empty_like(plain)
    _lib.check(lib.embnet_conv2d_patch_bnsums_f32(dyp.data_ptr(), L.weight_planes(wd, 1).data_ptr(), dx.data_ptr(), n, h, w, k, 3, 3, c, 1, 1,
                                                  h, w, ed.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                                  vec[3].data_ptr(), act, part.data_ptr(), rows, ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    assert torch.equal(dx, plain) and torch.isfinite(part).all()
    want = dgrad64(dy.astype(np.float64), wt.astype(np.float64), 1)
    z = e.astype(np.float64) * scale + shift
    dz = want * (z > 0) if act else want
    ehat = (e.astype(np.float64) - mean) * rstd
    got = part.double().sum(-1).cpu().numpy()
    assert np.abs(got[0] - dz.sum((0, 1, 2))).max() <= 1e-5 * np.abs(dz).sum((0, 1, 2)).max()
    assert np.abs(got[1] - (dz * ehat).sum((0, 1, 2))).max() <= 1e-5 * np.abs(dz * ehat).sum((0, 1, 2)).max()


def test_unit_backward_skips_the_batchnorm_reduction_pass(dev):
    """BatchNormalization(ReLU) -> patch conv: with the sums from the data gradient's epilogue the BatchNormalization backward
    starts at its finalize kernel (no bn_bwd_reduce launch for bn2), and every gradient equals the separate-pass chain within
    fp32 summation order."""
    gen = torch.Generator().manual_seed(5)
    net = _Pair(64, 96, gen).to(dev).train()
    x0 = torch.randn(6, 14, 14, 64, generator=gen).to(dev)
    dy = torch.randn(6, 14, 14, 64, generator=gen).to(dev)
    out, names = {}, {}
    default = L.PATCH_BN_SUMS[0]                  # (off: the epilogue costs more than the reduction launches it replaces)
    for on in (True, False):
        L.PATCH_BN_SUMS[0] = on
        try:
            x = x0.clone().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            y = net(x)
            _lib.trace_reset(); _lib.trace_enable(True)
            y.backward(dy)
            torch.cuda.synchronize()
            names[on] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            out[on] = [x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        finally:
            L.PATCH_BN_SUMS[0] = default
            L.BN_SUMS.clear()
    count = lambda on: sum("bn_bwd_reduce" in nm for nm in names[on])
    assert count(False) - count(True) >= 1, (names[True], names[False])
    for a, b in zip(out[True], out[False]):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 2e-5



def test_resnet18_patch_on_equals_off(dev):
    """Whole backbone, training mode: embeddings and every parameter gradient with the patch convs against the gather convs.
    A ReLU whose input sits within rounding of zero may take the other branch between the two runs (the fp32 accumulation
    order differs): one such flip moves single gradient tensors by ~1e-3, so the bulk is held tight and the rest loosely."""
    base, _ = B.get_backbone(input_shape=(64, 64, 3), encodings_len=64, backbone_name="resnet18", embeddings_normalization=True,
                             backbone_weights=None, seed=5)
    base.to(dev).train()
    imgs = torch.rand(16, 64, 64, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    g = torch.randn(16, 64, generator=torch.Generator().manual_seed(2)).to(dev)
    res = {}
    for on in (True, False):
        L.PATCH_CONV[0] = on
        try:
            for p in base.parameters():
                p.grad = None
            y = base(imgs)
            y.backward(g)
            torch.cuda.synchronize()
            res[on] = (y.detach().clone(), [p.grad.clone() for p in base.parameters()])
        finally:
            L.PATCH_CONV[0] = True
    assert not L.DY_PLANES
    assert float((res[True][0] - res[False][0]).norm() / res[False][0].norm()) < 2e-5
    rels = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(res[True][1], res[False][1])]
    assert max(rels) < 5e-2, max(rels)
    assert sum(r < 1e-4 for r in rels) >= 0.85 * len(rels), sorted(rels)[-12:]


def test_trainer_steps_with_patch_convs_graph_equals_eager(dev):
    """The captured step carries the kernel-planes refresh (layers.refresh_weight_planes) after the optimizer: replays must
    follow eager steps exactly, which they would not with stale planes."""
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import TripletTrainer
    out = {}
    for graph in (False, True):
        base, _ = B.get_backbone(input_shape=(64, 64, 3), encodings_len=64, backbone_name="resnet18", embeddings_normalization=True,
                                 backbone_weights=None, seed=5)
        base.to(dev)
        tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "adam", 1e-3), k_classes=4, k_samples=4, margin=0.5, graph=graph)
        gen = torch.Generator().manual_seed(1)
        losses = [float(tr.step(torch.rand(16, 64, 64, 3, generator=gen).to(dev))) for _ in range(12)]           # capture after GRAPH_WARMUP = 8
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        out[graph] = (losses, [p.detach().clone() for p in base.parameters()])
    assert out[True][0] == out[False][0], out
    assert all(torch.equal(a, b) for a, b in zip(out[True][1], out[False][1]))


def test_weight_planes_follow_the_weights(dev):
    """Planes are rebuilt after an in-place weight change (tensor version) and after KerasOptimizer.step (epoch)."""
    w = torch.randn(3, 3, 32, 64, device=dev)
    p0 = L.weight_planes(w, 0).clone()
    assert torch.equal(L.weight_planes(w, 0), p0)
    w.mul_(2.0)
    p1 = L.weight_planes(w, 0)
    torch.cuda.synchronize()
    assert not torch.equal(p1, p0)
    # forward layout [3][R][C/16][S][K][16]: decode and compare with the weights
    f = pieces(p1.cpu().numpy().view(np.uint16), (3, 2, 3, 64, 16))          # [r][cc][s][k][j]
    back = f.transpose(0, 2, 1, 4, 3).reshape(3, 3, 32, 64)                  # [r][s][cc*16+j][k]
    want = w.cpu().numpy().astype(np.float64)
    if F16:
        assert (np.abs(back - want) <= np.maximum(np.abs(want) * 2.0 ** -21, 2.0 ** -33)).all()      # (weights x 2^8: subnormal pieces below 2^-32)
    else:
        assert np.array_equal(back, want)


def test_three_piece_bf16_planes_in_a_child_process(dev):
    """The other planes format (EMBNET_PLANES_F16=0: three bf16 pieces, six products — the knob is read once per process): this
    file, the planes weight gradient's tests and the planes worst-case test run in a child pytest."""
    import subprocess
    import sys
    if not F16:
        pytest.skip("this process already runs the three-piece format")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMBNET_PLANES_F16="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_conv_patch_gpu.py", "tests/test_wgrad_planes_gpu.py",
                        "tests/test_round3_gpu.py", "-k", "not child_process and (patch or planes or decode or unit or resnet18_patch)"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
