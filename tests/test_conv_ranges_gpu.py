"""Three products per fp32 product on the implicit-GEMM ("gather") convolutions (csrc/conv.hip "Ranges", include/embnet.h
ABI 21: range slots are explicit arguments of the *_ex entry points), through the C ABI:

  * range slots: embnet_range_multi and the `dx_range` of embnet_bn_bwd_ex leave exactly the bit pattern of max |element|;
  * forward / data gradient / weight gradient of the layer classes the zoo ResNets keep on these kernels — the 7x7 stride-2
    stem, 3x3 stride 2, 1x1 stride 1 and 2 (reference embedding_net/backbones.py:99-104) — launched with both operands' ranges,
    against a float64 convolution of the same operands, beside the six-term kernels' error on the same data; gradients ten
    orders of magnitude below one and ACTIVATIONS of amplitude 1e-4 ... 1e5 (every operand has a scale: VERDICT r05 #1);
  * adversarial operands (all positive, every low mantissa bit set, K = 4608);
  * a ResNet18 training step: the ranges reach every gather conv (trace), nothing is left in the step context, and the loss
    matches the six-term build of the same step to fp32 rounding.
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


def bits(t):
    return int(t.view(torch.int32).item()) & 0xFFFFFFFF


def range_of(x):
    """Range slot of tensor x through embnet_range_multi."""
    lib = _lib.lib()
    slot = torch.full((1,), 0x12345678, dtype=torch.int32, device=x.device)       # (stale content: the call zeroes it)
    table = torch.tensor([[x.data_ptr(), x.numel(), slot.data_ptr()]], dtype=torch.int64, device=x.device)
    ce = lib.embnet_range_chunk_elems()
    chunks = torch.tensor([(0, j) for j in range(-(-x.numel() // ce))], dtype=torch.int32, device=x.device)
    _lib.check(lib.embnet_range_multi(table.data_ptr(), 1, chunks.data_ptr(), chunks.shape[0], _lib.stream()))
    return slot


def test_range_multi_is_the_exact_maximum(dev):
    g = torch.Generator().manual_seed(1)
    for n in (1, 5, 4095, 4096, 4097, 3 * 3 * 64 * 64, 7 * 7 * 3 * 64 + 3):
        x = (torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 3, (1,), generator=g))).to(dev)
        assert bits(range_of(x)) == bits(x.abs().max()), n
    z = torch.zeros(100, device=dev)
    assert bits(range_of(z)) == 0


def test_many_ranges_in_one_call(dev):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(2)
    xs = [torch.randn(n, generator=g).to(dev) * s for n, s in ((10, 1.0), (9000, 1e-3), (4096, 50.0), (1, 1e-9))]
    slots = torch.full((len(xs),), -1, dtype=torch.int32, device=dev)
    rows = [(x.data_ptr(), x.numel(), slots.data_ptr() + 4 * i) for i, x in enumerate(xs)]
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    ce = lib.embnet_range_chunk_elems()
    chunks = torch.tensor([(i, j) for i, x in enumerate(xs) for j in range(-(-x.numel() // ce))], dtype=torch.int32, device=dev)
    _lib.check(lib.embnet_range_multi(table.data_ptr(), len(xs), chunks.data_ptr(), chunks.shape[0], _lib.stream()))
    for i, x in enumerate(xs):
        assert bits(slots[i]) == bits(x.abs().max())


def test_bn_backward_emits_the_range_of_its_dx(dev):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for (m, c, scale, add) in ((128 * 49, 64, 1e-6, False), (5000, 256, 3.0, True), (37, 4, 1e-3, False)):
        x = torch.randn(m, c, generator=g).to(dev)
        dy = (torch.randn(m, c, generator=g) * scale).to(dev)
        dx_add = (torch.randn(m, c, generator=g) * scale).to(dev) if add else None
        mean, var = x.mean(0), x.var(0, unbiased=False)
        rstd = (var + 1e-3).rsqrt()
        gamma = torch.rand(c, generator=g).to(dev) + 0.5
        sc, sh = gamma * rstd, -mean * gamma * rstd
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ws = torch.empty(max(lib.embnet_bn_workspace_bytes(m, c) // 4, 4), device=dev)
        slot = torch.full((lib.embnet_range_slot_words(),), 0x7F000000, dtype=torch.int32, device=dev)   # (stale content: zeroed by the call)
        _lib.check(lib.embnet_bn_bwd_ex(dy.data_ptr(), x.data_ptr(), m, c, mean.data_ptr(), rstd.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                        1, 1, _lib.ptr(dx_add), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), None, ws.data_ptr(),
                                        ws.numel() * 4, slot.data_ptr(), None, None, _lib.stream()))
        assert bits(slot[0]) == bits(dx.abs().max())
        # the deprecated per-thread request (ABI 20) gives the same, once
        slot.fill_(0x7F000000)
        _lib.check(lib.embnet_range_emit(slot.data_ptr()))
        _lib.check(lib.embnet_bn_bwd(dy.data_ptr(), x.data_ptr(), m, c, mean.data_ptr(), rstd.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                     1, 1, _lib.ptr(dx_add), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), None, ws.data_ptr(),
                                     ws.numel() * 4, _lib.stream()))
        assert bits(slot[0]) == bits(dx.abs().max())
        # the request is consumed: a second call leaves the slot alone
        slot.fill_(7)
        _lib.check(lib.embnet_bn_bwd(dy.data_ptr(), x.data_ptr(), m, c, mean.data_ptr(), rstd.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                     1, 1, _lib.ptr(dx_add), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), None, ws.data_ptr(),
                                     ws.numel() * 4, _lib.stream()))
        assert bits(slot[0]) == 7 and bits(slot[-1]) == 7


def test_a_range_request_that_cannot_be_met_fails_loudly(dev):
    lib = _lib.lib()
    m, c = 64, 6                                         # c % 4 != 0: the scalar kernels, which emit no range
    x, dy = torch.randn(m, c, device=dev), torch.randn(m, c, device=dev)
    mean, rstd = x.mean(0), (x.var(0, unbiased=False) + 1e-3).rsqrt()
    one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    dx, dg, db = torch.empty_like(x), torch.empty(c, device=dev), torch.empty(c, device=dev)
    ws = torch.empty(max(lib.embnet_bn_workspace_bytes(m, c) // 4, 4), device=dev)
    slot = torch.zeros(lib.embnet_range_slot_words(), dtype=torch.int32, device=dev)
    rc = lib.embnet_bn_bwd_ex(dy.data_ptr(), x.data_ptr(), m, c, mean.data_ptr(), rstd.data_ptr(), one.data_ptr(), zero.data_ptr(), 0, 1,
                              None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), ws.numel() * 4, slot.data_ptr(), None, None,
                              _lib.stream())
    assert rc != 0 and b"range" in lib.embnet_last_error()
    _lib.check(lib.embnet_range_emit(slot.data_ptr()))
    rc = lib.embnet_bn_bwd(dy.data_ptr(), x.data_ptr(), m, c, mean.data_ptr(), rstd.data_ptr(), one.data_ptr(), zero.data_ptr(), 0, 1,
                           None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), ws.numel() * 4, _lib.stream())
    assert rc != 0 and b"range" in lib.embnet_last_error()
    # ... and the request is gone: the same call now succeeds
    _lib.check(lib.embnet_bn_bwd(dy.data_ptr(), x.data_ptr(), m, c, mean.data_ptr(), rstd.data_ptr(), one.data_ptr(), zero.data_ptr(), 0, 1,
                                 None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), ws.numel() * 4, _lib.stream()))


# ---- the three passes ---------------------------------------------------------------------------------------------------
def conv_fwd(x, w, geom, ranges=None):
    lib = _lib.lib()
    n, h, wd, c = x.shape
    r, s, _, k = w.shape
    stride, pt, pl, oh, ow = geom
    y = torch.full((n, oh, ow, k), float("nan"), device=x.device)
    ws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, c, r, s, k, oh, ow) // 4, 4), device=x.device)
    ra, rb = ranges if ranges is not None else (None, None)
    _lib.check(lib.embnet_conv2d_fwd_f32_ex(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, 0,
                                            None, None, None, 0, None, ws.data_ptr(), ws.numel() * 4, _lib.ptr(ra), _lib.ptr(rb), _lib.stream()))
    return y


def conv_dgrad(dy, w, xshape, geom, ranges=None):
    lib = _lib.lib()
    n, h, wd, c = xshape
    r, s, _, k = w.shape
    stride, pt, pl, oh, ow = geom
    dx = torch.full(xshape, float("nan"), device=dy.device)
    ws = torch.empty(max(lib.embnet_conv2d_dgrad_workspace_bytes(n, h, wd, c, r, s, k, stride) // 4, 4), device=dy.device)
    ra, rb = ranges if ranges is not None else (None, None)
    _lib.check(lib.embnet_conv2d_dgrad_f32_ex(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, 0,
                                              None, ws.data_ptr(), ws.numel() * 4, _lib.ptr(ra), _lib.ptr(rb), _lib.stream()))
    return dx


def conv_wgrad(x, dy, wshape, geom, ranges=None):
    lib = _lib.lib()
    n, h, wd, c = x.shape
    r, s, _, k = wshape
    stride, pt, pl, oh, ow = geom
    dw = torch.full(wshape, float("nan"), device=x.device)
    ws = torch.empty(max(lib.embnet_conv2d_wgrad_workspace_bytes(n, c, r, s, k, oh, ow) // 4, 4), device=x.device)
    ra, rb = ranges if ranges is not None else (None, None)
    _lib.check(lib.embnet_conv2d_wgrad_f32_ex(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, n, h, wd, c, r, s, k,
                                              stride, pt, pl, oh, ow, None, None, 0, _lib.ptr(ra), _lib.ptr(rb), _lib.stream()))
    return dw


def ref64(x, w, dy, stride, pad):
    """float64 forward, data gradient and weight gradient of y = conv(x [NHWC], w [RSCK]) on the CPU."""
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1).requires_grad_(True)
    y = torch.nn.functional.conv2d(xt, wt, stride=stride, padding=pad)
    y.backward(torch.from_numpy(dy.astype(np.float64)).permute(0, 3, 1, 2))
    return (y.detach().permute(0, 2, 3, 1).numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), wt.grad.permute(2, 3, 1, 0).numpy())


def rel(a, ref):
    return float(np.abs(a.astype(np.float64) - ref).max() / np.abs(ref).max())


LAYERS = [  # n, h, w, c, k, kernel, stride, pad, gradient magnitude
    (2, 64, 64, 4, 64, 7, 2, 3, 1e-4),        # stem (channel-padded image)
    (4, 28, 28, 64, 128, 3, 2, 1, 1e-7),      # stride-2 3x3 (basic units)
    (4, 28, 28, 64, 128, 1, 2, 0, 1e-7),      # its 1x1 projection shortcut
    (3, 14, 14, 256, 64, 1, 1, 0, 3e-9),      # bottleneck conv1
    (3, 14, 14, 64, 256, 1, 1, 0, 1e-2),      # bottleneck conv3
    (2, 9, 7, 512, 128, 1, 1, 0, 1e-5),       # ragged map, long reduction
    (1, 7, 7, 128, 512, 3, 2, 1, 1.0),
    (2, 16, 16, 64, 64, 3, 2, 1, 1e-3),       # 576 gradient rows x 64 filters: the weight gradient's 192-row tile
]


AMPLITUDES = [1.7, 1e-2, 1e-4, 3e4, 1e5]       # of the activation operand: O(1) was all round 5 tested (and all its scale-1 split could do)


@pytest.mark.parametrize("amp", AMPLITUDES)
@pytest.mark.parametrize("case", LAYERS)
def test_three_product_passes_vs_float64(dev, case, amp):
    n, h, wd, c, k, ks, stride, pad, gmag = case
    g = torch.Generator().manual_seed((hash(case) ^ int(amp * 7919)) & 0xFFFF)
    oh, ow = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    geom = (stride, pad, pad, oh, ow)
    x = torch.relu(torch.randn(n, h, wd, c, generator=g)) * amp                       # behind BatchNorm + ReLU: half zeros
    w = torch.randn(ks, ks, c, k, generator=g) * (2.0 / (ks * ks * c)) ** 0.5
    dy = torch.randn(n, oh, ow, k, generator=g) * gmag * torch.exp(2 * torch.randn(n, oh, ow, k, generator=g))   # log-normal spread
    y64, dx64, dw64 = ref64(x.numpy(), w.numpy(), dy.numpy(), stride, pad)
    xd, wd_, dyd = x.to(dev), w.to(dev), dy.to(dev)
    rw, rdy, rx = range_of(wd_), range_of(dyd), range_of(xd)
    # the activation's slot holds an UPPER BOUND in the network (the BatchNormalization's: up to a few binades loose): 8x here
    rx = (rx.view(torch.float32) * 8).view(torch.int32)
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        y3 = conv_fwd(xd, wd_, geom, (rx, rw))
        dx3 = conv_dgrad(dyd, wd_, tuple(x.shape), geom, (rdy, rw))
        dw3 = conv_wgrad(xd, dyd, tuple(w.shape), geom, (rx, rdy))
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert any("conv_fwd_h_kernel" in s for s in names) and any("conv_dgrad_h_kernel" in s for s in names) \
        and any("conv_wgrad_h_kernel" in s for s in names), names
    y6, dx6, dw6 = conv_fwd(xd, wd_, geom), conv_dgrad(dyd, wd_, tuple(x.shape), geom), conv_wgrad(xd, dyd, tuple(w.shape), geom)
    e3 = (rel(y3.cpu().numpy(), y64), rel(dx3.cpu().numpy(), dx64), rel(dw3.cpu().numpy(), dw64))
    e6 = (rel(y6.cpu().numpy(), y64), rel(dx6.cpu().numpy(), dx64), rel(dw6.cpu().numpy(), dw64))
    print(f"{case} amplitude {amp:g}: three products {e3[0]:.2e} {e3[1]:.2e} {e3[2]:.2e} | six terms {e6[0]:.2e} {e6[1]:.2e} {e6[2]:.2e}")
    # fp32-accumulation-sized errors (the six-term kernels' own are printed beside them), relative to the largest result element
    for a, b in zip(e3, e6):
        assert a < 1.5e-6 and a < 4 * b + 3e-7, (e3, e6)


def test_ranges_are_per_call(dev):
    """A conv call without ranges after one with them runs the six-term kernel (trace), bit-identical to a call before any
    range was ever given: nothing sticks to the thread."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 14, 14, 64, generator=g).to(dev)
    w = (torch.randn(1, 1, 64, 64, generator=g) * 0.1).to(dev)
    geom = (1, 0, 0, 14, 14)
    plain = conv_fwd(x, w, geom)
    rw, rx = range_of(w), range_of(x)
    conv_fwd(x, w, geom, (rx, rw))
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        again = conv_fwd(x, w, geom)
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert torch.equal(plain, again) and not any("_h_kernel" in s for s in names), names


def test_worst_case_operands(dev):
    """All-positive operands with every low mantissa bit set (no cancellation, the largest possible dropped terms), K = 4608:
    the measure of tests/test_round3_gpu.py::test_planes_split_worst_case on the in-kernel split."""
    g = torch.Generator().manual_seed(6)
    n, h, c, k = 2, 12, 512, 64

    def worst(shape, lo, hi):
        v = torch.rand(shape, generator=g) * (hi - lo) + lo
        b = v.view(torch.int32) | 0x00001FFF                        # the 13 bits below an fp16's mantissa all set
        return b.view(torch.float32)

    x = worst((n, h, h, c), 0.5, 2.0)
    w = worst((3, 3, c, k), 0.01, 0.04)
    geom = (2, 1, 1, 6, 6)
    dy = worst((n, 6, 6, k), 1e-6, 4e-6)
    y64, dx64, dw64 = ref64(x.numpy(), w.numpy(), dy.numpy(), 2, 1)
    xd, wd_, dyd = x.to(dev), w.to(dev), dy.to(dev)
    rw, rdy, rx = range_of(wd_), range_of(dyd), range_of(xd)
    e = (rel(conv_fwd(xd, wd_, geom, (rx, rw)).cpu().numpy(), y64),
         rel(conv_dgrad(dyd, wd_, tuple(x.shape), geom, (rdy, rw)).cpu().numpy(), dx64),
         rel(conv_wgrad(xd, dyd, tuple(w.shape), geom, (rx, rdy)).cpu().numpy(), dw64))
    print("worst case, three products: forward %.2e data gradient %.2e weight gradient %.2e" % e)
    assert max(e) < 1.2e-6, e


# ---- in the network -------------------------------------------------------------------------------------------------------
def _resnet18_step(dev, f16):
    from embeddingnet_amd import layers as L
    from embeddingnet_amd.backbones import get_backbone
    from embeddingnet_amd.train_step import TripletTrainer
    from embeddingnet_amd.optimizers import KerasOptimizer
    old = L.CONV_F16[0]
    L.CONV_F16[0] = f16
    try:
        torch.manual_seed(0)
        base, _ = get_backbone((64, 64, 3), encodings_len=64, backbone_name="resnet18", backbone_weights=None, seed=4, device=dev)
        base.train()
        tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "adam", 1e-3), k_classes=4, k_samples=4, margin=0.5,
                            negatives_selection_mode="hardest", graph=False)
        g = torch.Generator().manual_seed(9)
        losses = []
        _lib.trace_reset(); _lib.trace_enable(True)
        try:
            for i in range(3):
                x = torch.rand(16, 64, 64, 3, generator=g).to(dev)
                losses.append(float(tr.step(x)))
                if i == 0:
                    names = [r[0] for r in _lib.trace_records()]
        finally:
            _lib.trace_enable(False)
        left = tr.ctx.leftovers()
        return losses, names, left
    finally:
        L.CONV_F16[0] = old


def test_resnet18_step_runs_its_gather_convs_on_three_products(dev):
    l3, names3, left3 = _resnet18_step(dev, True)
    l6, names6, left6 = _resnet18_step(dev, False)
    assert not left3 and not left6, (left3, left6)
    h = [s for s in names3 if "_h_kernel" in s]
    six = [s for s in names3 if ("conv_fwd_kernel" in s or "conv_dgrad_kernel" in s or "conv_wgrad_kernel" in s)]
    assert not any("_h_kernel" in s for s in names6)
    # forward: stem (its own three-product kernel, csrc/conv_stem.hip) + 3 stride-2 3x3 + 4 shortcuts = 8 launches; backward: their
    # data / weight gradients (the stem has no data gradient)
    # (at 64x64 the last stages' maps are too small for the patch kernel: their 3x3 convs are gather launches too)
    assert sum("conv_fwd_h_kernel" in s for s in h) + sum("conv_stem_kernel" in s for s in names3) >= 8, (h, six)
    assert sum("conv_stem_kernel" in s for s in names3) == 1 and not any("conv_stem_kernel" in s for s in names6)
    assert sum("conv_wgrad_h_kernel" in s for s in h) >= 8 and sum("conv_dgrad_h_kernel" in s for s in h) >= 7, (h, six)
    assert not any("conv_fwd_kernel" in s for s in six), six
    # same start: the first two losses agree to fp32 rounding; from the third on two Adam updates (sign-like for gradients near
    # zero) separate the trajectories of ANY two arithmetics that differ in the last bits
    for i, (a, b) in enumerate(zip(l3, l6)):
        assert abs(a - b) <= (2e-5 if i < 2 else 2e-4) * max(abs(b), 1e-3), (l3, l6)


def test_optimizer_step_leaves_kernel_ranges_and_planes_current(dev):
    """KerasOptimizer.step() refreshes, in one launch each, the planes and ranges that its conv kernels have: the next forward
    launches no per-kernel refresh (trace), and every range slot holds the updated kernel's exact maximum."""
    from embeddingnet_amd import layers as L
    from embeddingnet_amd.backbones import get_backbone
    from embeddingnet_amd.optimizers import KerasOptimizer
    base, _ = get_backbone((64, 64, 3), encodings_len=32, backbone_name="resnet18", backbone_weights=None, seed=2, device=dev)
    base.train()
    opt = KerasOptimizer(base.parameters(), "sgd", 0.05)
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        x = torch.rand(8, 64, 64, 3, generator=g).to(dev)
        opt.zero_grad(set_to_none=True)
        base(x).square().sum().backward()
        opt.step()
    kernels = [m.kernel for m in base.modules() if isinstance(m, L.Conv2D)]
    ranged = [w for w in kernels if getattr(w, "_embnet_wrange", None) is not None]
    planed = [w for w in kernels if getattr(w, "_embnet_wplanes", None) is not None]
    assert len(ranged) >= 8
    for w in ranged:
        e = w._embnet_wrange
        assert e["epoch"] == L.WEIGHT_EPOCH[0] and e["version"] == w._version
        assert bits(e["slot"]) == bits(w.detach().abs().max())
    for w in planed:
        e = w._embnet_wplanes
        assert e["epoch"] == L.WEIGHT_EPOCH[0] and e["version"] == w._version
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        base(torch.rand(8, 64, 64, 3, generator=g).to(dev))
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert not any("range_multi" in s or "weight_planes" in s for s in names), [s for s in names if "range" in s or "planes" in s]
