"""Every operand of the two-piece fp16 conv arithmetic carries a RANGE — activations included (ABI 21).

Round 5 ran activations at scale 1: x = h1 + h2 with h2 an fp16 subnormal for every |x| < 2^-3, i.e. an absolute error of 2^-25
instead of a relative 2^-24 — 1.8e-4 of the result at activation amplitude 1e-4 — and a silent clamp above 65504.  Now the
training-mode BatchNormalization that writes an activation bounds max |y| from its statistics partials BEFORE its apply pass
(csrc/nn_kernels.hip channel_bound / tensor_bound), the planes take their scale from that bound and the gather convs read it as
the tensor's range slot; a tensor nobody vouches for runs the six-term bf16 kernels.  Through the C ABI:

  * the bound: never below the true maximum, within sqrt(rows per band) + the mean's share above it, for the statistics kernel's
    partials and for a conv epilogue's;
  * the planes a BatchNormalization writes decode to its fp32 output within the format's stated precision at output amplitudes
    1e-4 ... 1e5 (bound path and dry-run path);
  * BatchNorm -> planes -> patch conv (forward) and -> planes weight gradient against float64 at those amplitudes, same 1.5e-6 bound;
  * a conv call without either range runs six terms (trace); explicit ranges from two threads on two streams do not mix;
  * ResNet18 with every BatchNormalization's gamma at 1e-3 against the float64 oracle.
The gather kernels' own amplitude sweep is in tests/test_conv_ranges_gpu.py.  Reference layers: /root/reference/embedding_net/backbones.py:16,99-121.
"""
import threading

import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib
from embeddingnet_amd import layers as L

pytestmark = pytest.mark.gpu

AMPLITUDES = [1.7, 1e-2, 1e-4, 3e4, 1e5]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    if _lib.lib().embnet_conv_planes_mfma_terms() != 3:
        pytest.skip("the two-piece fp16 planes format is off (EMBNET_PLANES_F16=0)")
    return torch.device("cuda", 0)


def fbits(t):
    return float(t.view(torch.float32).item())


def bn_forward(x2d, gamma, beta, act, partial=None, y=True, planes=False, use_bound=True, eps=1e-3):
    """embnet_bn_train_fwd_ex (+ embnet_affine_act_planes_ex) on x2d [m, c] -> dict(y, bound [c], range (float), planes, stats)."""
    lib = _lib.lib()
    m, c = x2d.shape
    dev = x2d.device
    stats = torch.empty((7, c), device=dev)
    sp = stats.data_ptr()
    ws = torch.empty(max(lib.embnet_bn_workspace_bytes(m, c) // 4, 4), device=dev)
    yt = torch.full((m, c), float("nan"), device=dev) if y else None
    rng = torch.full((1,), -1, dtype=torch.int32, device=dev)
    _lib.check(lib.embnet_bn_train_fwd_ex(x2d.data_ptr(), m, c, _lib.ptr(gamma), _lib.ptr(beta), eps, 0.99, act,
                                          None if planes else _lib.ptr(yt), sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, None, None,
                                          _lib.ptr(partial), partial.shape[2] if partial is not None else 0, ws.data_ptr(), ws.numel() * 4,
                                          sp + 16 * c, rng.data_ptr() if (y and not planes) else None, sp + 24 * c, _lib.stream()))
    out = dict(y=yt, stats=stats, bound=stats[4], range=rng, xhat_bound=stats[6])
    if planes:
        p = torch.zeros(3 * m * c, dtype=torch.int16, device=dev)
        _lib.check(lib.embnet_affine_act_planes_ex(x2d.data_ptr(), m, c, sp + 8 * c, sp + 12 * c, act, _lib.ptr(yt), p.data_ptr(),
                                                   (sp + 16 * c) if use_bound else None, rng.data_ptr() if y else None, _lib.stream()))
        out["planes"] = p
    return out


def decode(planes, m, c):
    """two-piece fp16 planes [2+][c/16][m][16] -> (float64 [m, c], s)."""
    flat = planes.cpu().numpy().view(np.uint16).reshape(3, -1)
    s, inv = flat[2][:4].view(np.float32)[:2]
    assert s > 0 and s * inv == 1.0 and np.log2(s) == np.round(np.log2(s)), (s, inv)
    v = (flat[0].view(np.float16).astype(np.float64) + flat[1].view(np.float16).astype(np.float64)) * float(inv)
    return v.reshape(c // 16, m, 16).transpose(1, 0, 2).reshape(m, c), float(s)


# ---- the bound -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,c", [(128 * 49, 64), (5000, 256), (37, 16), (200000, 32)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_bound_from_the_statistics_kernel_is_sound_and_within_its_band_factor(dev, m, c, act):
    g = torch.Generator().manual_seed(m + c + act)
    # per-channel means up to 30 sigma, spreads over four decades, one heavy outlier
    x = torch.randn(m, c, generator=g) * torch.logspace(-2, 2, c) + torch.randn(c, generator=g) * 30 * torch.logspace(-2, 2, c)
    x[m // 3, c // 2] *= 40
    x = x.to(dev)
    gamma = (torch.randn(c, generator=g) * 2).to(dev)
    beta = torch.randn(c, generator=g).to(dev)
    o = bn_forward(x, gamma, beta, act)
    torch.cuda.synchronize()
    ymax = o["y"].abs().amax(0)
    assert torch.isfinite(o["bound"]).all()
    assert (o["bound"] >= ymax).all(), float((ymax - o["bound"]).max())
    assert fbits(o["range"]) == float(o["bound"].max())
    # looseness: sqrt(rows per statistics block) at most, + the mean's share (here means of up to 30 sigma: <= ~2^6 more)
    ratio = float(o["bound"].max() / ymax.max())
    print(f"m={m} c={c} act={act}: bound / max = {ratio:.1f}")
    assert ratio < 4096


def test_bound_from_a_conv_epilogue_is_sound_and_tight(dev):
    """The partials the ResNets' BatchNormalizations actually get: per 32 ... 96-row band from the producing conv's epilogue."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(11)
    for (n, h, c, k, ks) in ((8, 28, 64, 128, 3), (4, 14, 256, 64, 1), (16, 56, 64, 64, 3)):
        x = torch.relu(torch.randn(n, h, h, c, generator=g)).to(dev)
        w = (torch.randn(ks, ks, c, k, generator=g) * (2.0 / (ks * ks * c)) ** 0.5).to(dev)
        pad = ks // 2
        rows = lib.embnet_conv2d_fwd_stats_rows(n, c, ks, ks, k, h, h)
        assert rows > 0
        part = torch.empty((2, k, rows), device=dev)
        y = torch.empty((n, h, h, k), device=dev)
        ws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, c, ks, ks, k, h, h) // 4, 4), device=dev)
        _lib.check(lib.embnet_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), n, h, h, c, ks, ks, k, 1, pad, pad, h, h, 0, None,
                                             None, None, 0, part.data_ptr(), ws.data_ptr(), ws.numel() * 4, _lib.stream()))
        gamma = (torch.rand(k, generator=g) + 0.5).to(dev)
        beta = (torch.randn(k, generator=g) * 0.2).to(dev)
        o = bn_forward(y.view(-1, k), gamma, beta, 1, partial=part)
        torch.cuda.synchronize()
        ymax = o["y"].abs().amax(0)
        assert (o["bound"] >= ymax).all()
        ratio = float(o["bound"].max() / ymax.max())
        print(f"conv {ks}x{ks} {c}->{k} at {h}: bound / max = {ratio:.2f} ({rows} bands)")
        assert ratio < 16, ratio                       # <= sqrt(96) ~ 3.3 binades + the mean's share


# ---- the planes a BatchNormalization writes ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("amp", AMPLITUDES)
@pytest.mark.parametrize("use_bound", [True, False], ids=["bound", "dry-run"])
def test_batchnorm_planes_decode_to_the_fp32_output(dev, amp, use_bound):
    g = torch.Generator().manual_seed(int(amp * 1000) % 9973)
    m, c = 4 * 28 * 28, 64
    x = (torch.randn(m, c, generator=g) * 3 + 1).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev) * amp
    beta = (torch.randn(c, generator=g) * 0.3).to(dev) * amp
    o = bn_forward(x, gamma, beta, 1, planes=True, use_bound=use_bound)
    torch.cuda.synchronize()
    y = o["y"].cpu().numpy().astype(np.float64)
    got, s = decode(o["planes"], m, c)
    B = fbits(o["range"])
    top = np.abs(y).max()
    assert B >= top and top * s < 2.0 ** 15
    if use_bound:
        assert 2.0 ** 14 <= B * s < 2.0 ** 15            # the scale puts the BOUND into [2^14, 2^15) ...
        assert top * s >= 2.0 ** 10                      # ... and the bound is within a few binades of the maximum
    else:
        assert B == np.float32(top)                      # the dry run's exact maximum
        assert 2.0 ** 14 <= top * s < 2.0 ** 15
    # the format's stated precision (gemm_engine.h PRECISION): 2^-22 relative, or the subnormal floor 2^-25 / s absolute
    err = np.abs(got - y)
    tol = np.maximum(np.abs(y) * 2.0 ** -21, 2.0 ** -25 / s)
    assert (err <= tol).all(), (float(err.max()), float(top), s)
    # ... which at any amplitude is far below fp32's own resolution of the tensor's typical element
    rms = np.sqrt((y[y > 0] ** 2).mean())
    assert 2.0 ** -25 / s < rms * 2.0 ** -24, (s, rms)


def conv64(x, w, pad):
    y = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), padding=pad)
    return y.permute(0, 2, 3, 1).numpy()


def wgrad64(x, dy):
    xt = torch.from_numpy(x).permute(3, 0, 1, 2)
    dt = torch.from_numpy(dy).permute(3, 0, 1, 2)
    return torch.nn.functional.conv2d(xt, dt, padding=1).permute(2, 3, 0, 1).numpy()


@pytest.mark.parametrize("amp", AMPLITUDES)
def test_batchnorm_planes_through_patch_conv_and_planes_weight_gradient_vs_float64(dev, amp):
    """BatchNormalization(gamma ~ amp) + ReLU -> planes -> 3x3 patch conv forward and planes weight gradient, against float64
    convolutions of the BatchNormalization's own fp32 output: the error bound of the O(1) tests at every amplitude."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3 + int(np.log10(amp) * 10))
    n, h, c, k = 4, 28, 64, 64
    if not (lib.embnet_conv2d_patch_supported(n, c, 3, 3, k, 1, h, h) and lib.embnet_conv2d_wgrad_planes_supported(n, h, h, c, 3, 3, k, 1, 1, 1, h, h)):
        pytest.skip("geometry not served by the planes kernels")
    x = (torch.randn(n * h * h, c, generator=g) * 2 + 0.5).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev) * amp
    beta = (torch.randn(c, generator=g) * 0.2).to(dev) * amp
    o = bn_forward(x, gamma, beta, 1, planes=True)
    a = o["y"].view(n, h, h, c)
    w = (torch.randn(3, 3, c, k, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev)
    y = torch.empty((n, h, h, k), device=dev)
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(n, c, 3, 3, k, h, h), 4) // 4, device=dev)
    _lib.check(lib.embnet_conv2d_patch_f32(o["planes"].data_ptr(), L.weight_planes(w, 0).data_ptr(), None, y.data_ptr(), n, h, h, c, 3, 3, k,
                                           1, 1, h, h, 0, None, None, ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    want = conv64(a.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), 1)
    e_fwd = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    # weight gradient: dy as planes from an fp32 tensor (its own abs-max pass)
    dy = (torch.randn(n, h, h, k, generator=g) * 1e-3).to(dev)
    dyp = torch.empty(3 * dy.numel(), dtype=torch.int16, device=dev)
    _lib.check(lib.embnet_planes_from_f32(dy.data_ptr(), dy.numel() // k, k, dyp.data_ptr(), _lib.stream()))
    dw = torch.empty((3, 3, c, k), device=dev)
    ws2 = torch.empty(max(lib.embnet_conv2d_wgrad_planes_workspace_bytes(n, h, h, c, k) // 4, 4), device=dev)
    _lib.check(lib.embnet_conv2d_wgrad_planes_f32(o["planes"].data_ptr(), dyp.data_ptr(), dw.data_ptr(), ws2.data_ptr(), ws2.numel() * 4,
                                                  n, h, h, c, k, 1, _lib.stream()))
    wantw = wgrad64(a.cpu().numpy().astype(np.float64), dy.cpu().numpy().astype(np.float64))
    e_wg = np.abs(dw.cpu().numpy() - wantw).max() / np.abs(wantw).max()
    print(f"amplitude {amp:g}: forward {e_fwd:.2e}, weight gradient {e_wg:.2e}")
    assert e_fwd < 1.5e-6 and e_wg < 1.5e-6, (e_fwd, e_wg)


# ---- the BatchNorm backward's planes without a dry run -----------------------------------------------------------------
@pytest.mark.parametrize("m,c,gmag,add", [(8 * 28 * 28, 64, 1e-3, False), (4 * 14 * 14, 256, 1e-8, True), (37, 16, 3.0, True),
                                          (16 * 56 * 56, 64, 1e-5, True)])
def test_bn_backward_planes_take_their_scale_from_a_bound(dev, m, c, gmag, add):
    """dx as planes (+ fp32) with the scale from |scale| (max |dz| + |dbeta| / m + max |xhat| |dgamma| / m) [+ the range of dx_add]:
    the bound is never below max |dx| and within 4x of it, the planes decode to the fp32 dx, the fp32 dx and the parameter gradients
    are bit for bit those of the dry-run form, and the trace shows ONE apply pass."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(m + c)
    x = (torch.randn(m, c, generator=g) * torch.logspace(-1, 1, c) + 0.3).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    beta = (torch.randn(c, generator=g) * 0.1).to(dev)
    fwd = bn_forward(x, gamma, beta, 1)
    st, sp = fwd["stats"], fwd["stats"].data_ptr()
    dy = (torch.randn(m, c, generator=g) * gmag * torch.exp(torch.randn(m, c, generator=g))).to(dev)
    dx_add = (torch.randn(m, c, generator=g) * gmag * 3).to(dev) if add else None
    add_range = range_of(dx_add) if add else None
    ws = torch.empty(max(lib.embnet_bn_workspace_bytes(m, c) // 4, 4), device=dev)

    def run(xhat, addr):
        dx = torch.full((m, c), float("nan"), device=dev)
        planes = torch.zeros(3 * m * c, dtype=torch.int16, device=dev)
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        slot = torch.full((lib.embnet_range_slot_words(),), -1, dtype=torch.int32, device=dev)
        _lib.trace_reset(); _lib.trace_enable(True)
        try:
            _lib.check(lib.embnet_bn_bwd_ex(dy.data_ptr(), x.data_ptr(), m, c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, 1, 1, _lib.ptr(dx_add),
                                            dx.data_ptr(), dg.data_ptr(), db.data_ptr(), planes.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                            slot.data_ptr(), xhat, addr, _lib.stream()))
            names = [r[0] for r in _lib.trace_records()]
        finally:
            _lib.trace_enable(False)
        return dx, planes, dg, db, slot, names

    dx, planes, dg, db, slot, names = run(sp + 24 * c, _lib.ptr(add_range))
    dx0, planes0, dg0, db0, slot0, names0 = run(None, None)                       # no bound at hand: the dry run
    assert any("bn_bwd_apply4_kernel<4>" in s for s in names) and not any("bn_bwd_apply4_kernel<1>" in s for s in names), names
    assert any("bn_bwd_apply4_kernel<1>" in s for s in names0), names0
    assert torch.equal(dx, dx0) and torch.equal(dg, dg0) and torch.equal(db, db0)
    top = float(dx.abs().max())
    B, B0 = fbits(slot[:1]), fbits(slot0[:1])
    assert B0 == np.float32(top)                                                # the dry run's exact maximum
    assert top <= B <= 4 * top, (top, B)
    got, s = decode(planes, m, c)
    want = dx.cpu().numpy().astype(np.float64)
    assert 2.0 ** 14 <= B * s < 2.0 ** 15
    assert (np.abs(got - want) <= np.maximum(np.abs(want) * 2.0 ** -21, 2.0 ** -25 / s)).all()


def test_data_gradient_epilogue_hands_the_bound_its_third_plane(dev):
    """embnet_conv2d_dgrad_bnsums_f32_ex writes [3][c][rows]: the two BatchNorm-backward sums and max |dz| per row band;
    embnet_bn_bwd_partials_ex (partial_kinds = 3) then scales its dx planes without a dry run."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(21)
    n, h, c, k = 8, 14, 64, 256                                                  # a bottleneck's conv3 (1x1) behind bn3
    m = n * h * h
    e = (torch.randn(m, c, generator=g) + 0.2).to(dev)                          # bn3's input
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.1).to(dev)
    fwd = bn_forward(e, gamma, beta, 1)
    sp = fwd["stats"].data_ptr()
    w = (torch.randn(1, 1, c, k, generator=g) * 0.1).to(dev)
    dy = (torch.randn(n, h, h, k, generator=g) * 1e-4).to(dev)
    rows = lib.embnet_conv2d_dgrad_bnsums_rows(n, h, h, c, 1, 1, k, 1)
    assert rows > 0
    part = torch.full((3, c, rows), float("nan"), device=dev)
    da = torch.empty((n, h, h, c), device=dev)                                   # d(conv3 input) = d(act(bn3(e)))
    ws = torch.empty(max(lib.embnet_conv2d_dgrad_workspace_bytes(n, h, h, c, 1, 1, k, 1) // 4, 4), device=dev)
    _lib.check(lib.embnet_conv2d_dgrad_bnsums_f32_ex(dy.data_ptr(), w.data_ptr(), da.data_ptr(), n, h, h, c, 1, 1, k, 1, 0, 0, h, h,
                                                     e.data_ptr(), sp + 8 * c, sp + 12 * c, sp, sp + 4 * c, 1, part.data_ptr(), rows,
                                                     ws.data_ptr(), ws.numel() * 4, None, None, _lib.stream()))
    torch.cuda.synchronize()
    z = e * fwd["stats"][2] + fwd["stats"][3]
    dz = torch.where(z > 0, da.view(m, c), torch.zeros_like(z))
    assert torch.isfinite(part).all()
    assert float(part[2].amax()) == float(dz.abs().max())                        # the bands' maxima cover every element, exactly
    dx = torch.empty((m, c), device=dev)
    planes = torch.zeros(3 * m * c, dtype=torch.int16, device=dev)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        _lib.check(lib.embnet_bn_bwd_partials_ex(da.data_ptr(), e.data_ptr(), m, c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, 1, part.data_ptr(), rows,
                                                 None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), planes.data_ptr(), None, 3, sp + 24 * c, None,
                                                 _lib.stream()))
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert any("bn_bwd_apply4_kernel<4>" in s for s in names) and not any("bn_bwd_apply4_kernel<1>" in s for s in names), names
    got, s = decode(planes, m, c)
    want = dx.cpu().numpy().astype(np.float64)
    top = np.abs(want).max()
    assert 2.0 ** 12 <= top * s < 2.0 ** 15                                      # the bound is within 4x of the maximum
    assert (np.abs(got - want) <= np.maximum(np.abs(want) * 2.0 ** -21, 2.0 ** -25 / s)).all()


# ---- the C ABI's explicit ranges ---------------------------------------------------------------------------------------------------
def range_of(x):
    lib = _lib.lib()
    slot = torch.zeros(1, dtype=torch.int32, device=x.device)
    table = torch.tensor([[x.data_ptr(), x.numel(), slot.data_ptr()]], dtype=torch.int64, device=x.device)
    ce = lib.embnet_range_chunk_elems()
    chunks = torch.tensor([(0, j) for j in range(-(-x.numel() // ce))], dtype=torch.int32, device=x.device)
    _lib.check(lib.embnet_range_multi(table.data_ptr(), 1, chunks.data_ptr(), chunks.shape[0], _lib.stream()))
    return slot


def fwd_ex(x, w, xr, wr, stream=None):
    lib = _lib.lib()
    n, h, wd, c = x.shape
    k = w.shape[-1]
    y = torch.empty((n, h, wd, k), device=x.device)
    _lib.check(lib.embnet_conv2d_fwd_f32_ex(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), n, h, wd, c, 1, 1, k, 1, 0, 0, h, wd, 0, None, None,
                                            None, 0, None, None, 0, _lib.ptr(xr), _lib.ptr(wr), stream if stream is not None else _lib.stream()))
    return y


def test_a_call_without_both_ranges_runs_six_terms(dev):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 14, 14, 64, generator=g) * 1e-4).to(dev)
    w = (torch.randn(1, 1, 64, 64, generator=g) * 0.1).to(dev)
    rx, rw = range_of(x), range_of(w)
    plain = fwd_ex(x, w, None, None)
    for xr, wr, h in ((None, rw, False), (rx, None, False), (rx, rw, True)):
        _lib.trace_reset(); _lib.trace_enable(True)
        try:
            y = fwd_ex(x, w, xr, wr)
            names = [r[0] for r in _lib.trace_records()]
        finally:
            _lib.trace_enable(False)
        assert any("_h_kernel" in s for s in names) == h, (names, h)
        if not h:
            assert torch.equal(y, plain)
    # the deprecated per-thread request still works — and a NULL operand in it means "unknown" now, not "scale 1"
    lib = _lib.lib()
    _lib.check(lib.embnet_conv2d_ranges(None, rw.data_ptr()))
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        y = fwd_ex(x, w, None, None)                   # (an _ex call clears the armed request)
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert not any("_h_kernel" in s for s in names) and torch.equal(y, plain)


def test_two_threads_on_two_streams_keep_their_own_ranges(dev):
    """Explicit arguments: nothing a thread does between another thread's calls can change that thread's arithmetic."""
    g = torch.Generator().manual_seed(8)
    xs = [(torch.randn(2, 14, 14, 64, generator=g) * a).to(dev) for a in (1e-4, 3e3)]
    ws = [(torch.randn(1, 1, 64, 64, generator=g) * b).to(dev) for b in (0.1, 1e-5)]
    rs = [(range_of(x), range_of(w)) for x, w in zip(xs, ws)]
    torch.cuda.synchronize()
    alone = [fwd_ex(x, w, r[0], r[1]) for x, w, r in zip(xs, ws, rs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    out = [[None] * 20 for _ in range(2)]
    go = threading.Barrier(2)

    def work(i):
        torch.cuda.set_device(dev)
        go.wait()
        with torch.cuda.stream(streams[i]):
            for j in range(20):
                out[i][j] = fwd_ex(xs[i], ws[i], rs[i][0], rs[i][1], streams[i].cuda_stream)
        streams[i].synchronize()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        for j in range(20):
            assert torch.equal(out[i][j], alone[i]), (i, j)


# ---- in the network ------------------------------------------------------------------------------------------------------------------
def test_resnet18_with_small_gammas_vs_oracle(dev):
    """Every BatchNormalization's gamma at 1e-3 (and beta at +-2e-4): activations of amplitude 1e-3 everywhere — zero-init-residual /
    fine-tuned weights territory (reference backbones.py:16: pretrained weights are the default).  Embeddings and gradients against the
    float64 oracle, the bounds of tests/test_backbone_gpu.py; and the step runs the three-product kernels (trace)."""
    from embeddingnet_amd import backbones as B
    from oracle import backbones as OB
    shape, enc, batch = (64, 64, 3), 64, 8
    base, _ = B.get_backbone(shape, encodings_len=enc, backbone_name="resnet18", backbone_weights=None, seed=1, device=dev)
    rs = np.random.RandomState(0)
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, L.BatchNormalization):
                if m.gamma is not None:
                    m.gamma.fill_(1e-3)
                m.beta.copy_(torch.from_numpy((rs.randn(m.beta.numel()) * 2e-4).astype(np.float32)))
    L.WEIGHT_EPOCH[0] += 1
    x = rs.rand(batch, *shape).astype(np.float32)
    base.train()
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        emb = base(torch.from_numpy(x).to(dev))
        wgt = rs.randn(batch, enc).astype(np.float32)
        (emb * torch.from_numpy(wgt).to(dev)).sum().backward()
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert sum("_h_kernel" in s or "conv_stem_kernel" in s for s in names) >= 20, names
    assert not any(s.startswith("void embnet::conv_fwd_kernel") for s in names), [s for s in names if "conv_fwd_kernel" in s]
    params = {k: v.detach().cpu().double().requires_grad_(v.requires_grad) for k, v in B.keras_weights(base).items()}
    ctx = OB.Ctx(params, training=True)
    embr = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name="resnet18", encodings_len=enc)
    e = float((emb.detach().cpu().double() - embr.detach()).abs().max() / embr.detach().abs().max())
    assert e < 2e-4, e
    (embr * torch.tensor(wgt, dtype=torch.float64)).sum().backward()
    got = B.keras_weights(base)
    num = den = 0.0
    for k, p in ctx.params.items():
        if p.grad is None:
            continue
        diff = got[k].grad.detach().cpu().double() - p.grad
        num += float((diff ** 2).sum()); den += float((p.grad ** 2).sum())
        assert float(diff.abs().max()) <= 0.3 * max(float(p.grad.abs().max()), 1e-12), k
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5
