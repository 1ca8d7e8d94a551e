"""The small backbones (`simple`, `simple2`: reference backbones.py:19-81) on three products per fp32 product.

`simple` has no BatchNormalization to bound its activations and its gradients come out of ReLU / MaxPool backward passes, so the
ranges the three-product kernels need are EXACT maxima that ride on passes which touch every element anyway (ABI 22):
  embnet_pad_channels_ex (the image), embnet_maxpool_fwd_ex (pooled activations), embnet_maxpool_relu_bwd_colsum_ex /
  embnet_relu_bwd_colsum_ex / embnet_bn_bwd_inrelu[_dropout]_ex (the gradient behind a fused ReLU's mask).
Here: each `_ex` pass returns the plain call's values bit for bit and the exact range; the backbones' traces hold no six-term conv
kernel where c and k are multiples of four; activations and parameter gradients agree with the float64 oracle at input amplitudes
1 and 1e-3 (a fixed scale of 1 — round 5's arithmetic — fails the second).
"""
import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib
from embeddingnet_amd._lib import check, ptr, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def bits(t):
    return int(t.detach().reshape(-1)[:1].view(torch.int32).item()) & 0xFFFFFFFF


def fbits(v):
    return int(np.float32(v).view(np.uint32))


def slot(dev):
    return torch.full((_lib.lib().embnet_range_slot_words(),), -1, dtype=torch.int32, device=dev)     # (garbage in: the call zeroes it)


def test_pad_channels_ex(dev):
    lib = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.rand((5, 9, 7, 3), device=dev, generator=g) - 0.3) * 3e-3
    a, b = torch.empty((5, 9, 7, 4), device=dev), torch.empty((5, 9, 7, 4), device=dev)
    s = slot(dev)
    check(lib.embnet_pad_channels(ptr(x), 5 * 9 * 7, 3, 4, ptr(a), stream()))
    check(lib.embnet_pad_channels_ex(ptr(x), 5 * 9 * 7, 3, 4, ptr(b), ptr(s), stream()))
    assert torch.equal(a, b) and torch.equal(a[..., :3], x) and float(a[..., 3].abs().max()) == 0.0
    assert bits(s) == fbits(float(x.abs().max()))


@pytest.mark.parametrize("shape,k,stride,pad", [((3, 12, 10, 8), 2, 2, 0), ((2, 15, 15, 64), 3, 2, 1), ((33, 48, 48, 64), 2, 2, 0)])
def test_maxpool_fwd_ex(dev, shape, k, stride, pad):
    lib = _lib.lib()
    n, h, w, c = shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    g = torch.Generator(device=dev).manual_seed(2)
    x = (torch.rand(shape, device=dev, generator=g) - 0.6) * 7.0
    ya, yb = torch.empty((n, oh, ow, c), device=dev), torch.empty((n, oh, ow, c), device=dev)
    aa, ab = torch.empty((n, oh, ow, c), device=dev, dtype=torch.uint8), torch.empty((n, oh, ow, c), device=dev, dtype=torch.uint8)
    s = slot(dev)
    check(lib.embnet_maxpool_fwd(ptr(x), n, h, w, c, k, stride, pad, oh, ow, ptr(ya), ptr(aa), stream()))
    check(lib.embnet_maxpool_fwd_ex(ptr(x), n, h, w, c, k, stride, pad, oh, ow, ptr(yb), ptr(ab), ptr(s), stream()))
    assert torch.equal(ya, yb) and torch.equal(aa, ab)
    assert bits(s) == fbits(float(ya.abs().max()))


def test_maxpool_fwd_ex_refuses_a_range_for_odd_channel_counts(dev):
    lib = _lib.lib()
    x = torch.rand((1, 4, 4, 3), device=dev)
    y = torch.empty((1, 2, 2, 3), device=dev)
    a = torch.empty((1, 2, 2, 3), device=dev, dtype=torch.uint8)
    assert lib.embnet_maxpool_fwd_ex(ptr(x), 1, 4, 4, 3, 2, 2, 0, 2, 2, ptr(y), ptr(a), ptr(slot(dev)), stream()) != 0
    assert b"c % 4" in lib.embnet_last_error()


def test_maxpool_relu_bwd_colsum_ex(dev):
    lib = _lib.lib()
    n, h, w, c, k, st = 6, 20, 18, 32, 2, 2
    oh, ow = h // 2, w // 2
    g = torch.Generator(device=dev).manual_seed(3)
    y = torch.relu(torch.randn((n, h, w, c), device=dev, generator=g))
    p = torch.empty((n, oh, ow, c), device=dev)
    arg = torch.empty((n, oh, ow, c), device=dev, dtype=torch.uint8)
    check(lib.embnet_maxpool_fwd(ptr(y), n, h, w, c, k, st, 0, oh, ow, ptr(p), ptr(arg), stream()))
    dy = torch.randn((n, oh, ow, c), device=dev, generator=g) * 2e-5
    ws = torch.empty(lib.embnet_bn_workspace_bytes(n * h * w, c) // 4 + 4, device=dev)
    outs = []
    s = slot(dev)
    for rng in (None, s):
        dz, db = torch.empty_like(y), torch.empty(c, device=dev)
        check(lib.embnet_maxpool_relu_bwd_colsum_ex(ptr(dy), ptr(arg), ptr(y), n, h, w, c, k, st, 0, oh, ow, ptr(dz), ptr(db), ptr(ws),
                                                    ws.numel() * 4, ptr(rng), stream()))
        outs.append((dz, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bits(s) == fbits(float(outs[0][0].abs().max())) and float(outs[0][0].abs().max()) > 0


def test_relu_bwd_colsum_ex(dev):
    lib = _lib.lib()
    m, c = 1152, 256
    g = torch.Generator(device=dev).manual_seed(4)
    y = torch.relu(torch.randn((m, c), device=dev, generator=g))
    dy = torch.randn((m, c), device=dev, generator=g) * 3e4
    ws = torch.empty(lib.embnet_colsum_workspace_bytes(m, c) // 4 + 4, device=dev)
    s = slot(dev)
    dz0, db0, dz1, db1 = torch.empty_like(y), torch.empty(c, device=dev), torch.empty_like(y), torch.empty(c, device=dev)
    check(lib.embnet_relu_bwd_colsum(ptr(dy), ptr(y), m, c, ptr(dz0), ptr(db0), ptr(ws), ws.numel() * 4, stream()))
    check(lib.embnet_relu_bwd_colsum_ex(ptr(dy), ptr(y), m, c, ptr(dz1), ptr(db1), ptr(ws), ws.numel() * 4, ptr(s), stream()))
    assert torch.equal(dz0, dz1) and torch.equal(db0, db1) and torch.equal(dz0, dy * (y > 0))
    assert bits(s) == fbits(float(dz0.abs().max()))


@pytest.mark.parametrize("rate", [0.0, 0.4])
def test_bn_bwd_inrelu_ex(dev, rate):
    lib = _lib.lib()
    m, c = 32 * 30 * 30, 32
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.relu(torch.randn((m, c), device=dev, generator=g) * 0.7)
    dy = torch.randn((m, c), device=dev, generator=g) * 1e-3
    gamma, beta = torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g) * 0.1
    stats = torch.empty((4, c), device=dev)
    mm, mv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    ws = torch.empty(lib.embnet_bn_workspace_bytes(m, c) // 4 + 4, device=dev)
    sp = stats.data_ptr()
    check(lib.embnet_bn_train_fwd(ptr(x), m, c, ptr(gamma), ptr(beta), 1e-3, 0.99, 0, None, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c,
                                  ptr(mm), ptr(mv), None, 0, ptr(ws), ws.numel() * 4, stream()))
    s = slot(dev)
    outs = []
    for rng in (None, s):
        dz, dg, db, dbias = torch.empty_like(x), torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(c, device=dev)
        if rate:
            check(lib.embnet_bn_bwd_inrelu_dropout_ex(ptr(dy), ptr(x), m, c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, 0, 1, rate, 77, None,
                                                      ptr(dz), ptr(dg), ptr(db), ptr(dbias), ptr(ws), ws.numel() * 4, ptr(rng), stream()))
        else:
            check(lib.embnet_bn_bwd_inrelu_ex(ptr(dy), ptr(x), m, c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, 0, 1,
                                              ptr(dz), ptr(dg), ptr(db), ptr(dbias), ptr(ws), ws.numel() * 4, ptr(rng), stream()))
        outs.append((dz, dg, db, dbias))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert bits(s) == fbits(float(outs[0][0].abs().max())) and float(outs[0][0].abs().max()) > 0


# ---- in the backbones ------------------------------------------------------------------------------------------------------
SHAPES = {"simple": (105, 105, 3), "simple2": (64, 64, 3)}


def _base(dev, name, seed=3):
    from embeddingnet_amd.backbones import get_backbone
    torch.manual_seed(0)
    base, _ = get_backbone(SHAPES[name], encodings_len=64, backbone_name=name, backbone_weights=None, seed=seed, device=dev)
    base.train()
    return base


def _conv_launches(dev, name, f16):
    from embeddingnet_amd import layers as L
    old = L.CONV_F16[0]
    L.CONV_F16[0] = f16
    try:
        base = _base(dev, name)
        g = torch.Generator().manual_seed(1)
        x = torch.rand((8,) + SHAPES[name], generator=g).to(dev)
        t = torch.randn((8, 64), generator=g).to(dev)    # (the embeddings are L2-normalised: a sum of squares would have no gradient)
        ctx = L.current_context()
        (base(x) * t).sum().backward()                 # (one untraced step: kernel ranges, workspaces)
        ctx.unclaimed.clear()
        for p in base.parameters():
            p.grad = None
        _lib.trace_reset(); _lib.trace_enable(True)
        try:
            (base(x) * t).sum().backward()
            names = [r[0] for r in _lib.trace_records()]
        finally:
            _lib.trace_enable(False)
        left = dict(ctx.leftovers())
        left.update({"unclaimed " + k: v for k, v in ctx.unclaimed.items() if v})      # (every range a backward pass left was claimed)
        return names, left, [p.grad.clone() for p in base.parameters() if p.grad is not None]
    finally:
        L.CONV_F16[0] = old


def _six(names):
    return [s for s in names if any(t in s for t in ("conv_fwd_kernel", "conv_dgrad_kernel", "conv_wgrad_kernel"))]


def _three(names):
    return [s for s in names if "_h_kernel" in s]


def test_simple_runs_every_conv_pass_on_three_products(dev):
    names, left, g3 = _conv_launches(dev, "simple", True)
    assert not left, left
    # four convs: 4 forward, 3 data gradients (the image takes none), 4 weight gradients — the 3-channel image is widened to four
    # channels by the pass that also leaves its range
    assert not _six(names), _six(names)
    h = _three(names)
    assert sum("conv_fwd_h" in s for s in h) == 4 and sum("conv_dgrad_h" in s for s in h) == 3 and sum("conv_wgrad_h" in s for s in h) == 4, h
    names6, left6, g6 = _conv_launches(dev, "simple", False)
    assert not left6 and not _three(names6) and len(_six(names6)) == 11
    for a, b in zip(g3, g6):                       # two fp32-exact arithmetics: the gradients agree to fp32 rounding
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-12


def test_simple2_runs_its_4_multiple_convs_on_three_products(dev):
    names, left, g3 = _conv_launches(dev, "simple2", True)
    assert not left, left
    # conv1 reads the 3-channel image (scalar gathers: six terms, one forward + one weight gradient); conv2 ... conv7: 6 forward,
    # 6 data gradients, 6 weight gradients on three products
    six, h = _six(names), _three(names)
    assert sum("conv_fwd_h" in s for s in h) == 6 and sum("conv_dgrad_h" in s for s in h) == 6 and sum("conv_wgrad_h" in s for s in h) == 6, (h, six)
    assert len(six) == 2, six
    names6, left6, g6 = _conv_launches(dev, "simple2", False)
    assert not left6 and not _three(names6)
    for a, b in zip(g3, g6):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12


@pytest.mark.parametrize("name", ["simple", "simple2"])
@pytest.mark.parametrize("amplitude", [1.0, 1e-3])
def test_small_backbone_vs_float64_oracle_at_small_input_amplitude(dev, name, amplitude):
    """Embeddings and parameter gradients of one training-mode pass against the float64 oracle, with the image scaled by
    `amplitude`: `simple` has no normalisation, so every activation (and the kernel gradients) scales with it — the regime in which
    a fixed operand scale of 1 put the fp16 low pieces on their subnormal floor.  Same acceptance as
    tests/test_backbone_gpu.py::test_backbone_forward_backward_vs_oracle: every gradient tensor within 5x the float32 oracle's own
    deviation from the float64 oracle (+ 1e-4), one tensor in forty excused (a ReLU / arg-max decision that fell the other way)."""
    from embeddingnet_amd import backbones as B
    from oracle import backbones as OB
    from tests.test_backbone_gpu import _oracle_from
    enc, batch = 64, 6
    base, _ = B.get_backbone(SHAPES[name], encodings_len=enc, backbone_name=name, backbone_weights=None, seed=5, device=dev)
    for m in base.modules():
        if hasattr(m, "enabled"):
            m.enabled = False                             # dropout off for parity
    base.train()
    rs = np.random.RandomState(11)
    x = (rs.rand(batch, *SHAPES[name]) * amplitude).astype(np.float32)
    wgt = rs.randn(batch, enc).astype(np.float32)
    emb = base(torch.tensor(x).to(dev))
    (emb * torch.tensor(wgt).to(dev)).sum().backward()
    ctx = _oracle_from(base, training=True)
    embr = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc)
    (embr * torch.tensor(wgt, dtype=torch.float64)).sum().backward()
    ctx32 = _oracle_from(base, training=True, dtype=torch.float32)
    emb32 = OB.base_model(ctx32, torch.tensor(x), backbone_name=name, encodings_len=enc)
    (emb32 * torch.tensor(wgt)).sum().backward()
    scale = float(embr.abs().max())
    floor = float((emb32.double() - embr).abs().max()) / scale
    err = float((emb.detach().cpu().double() - embr.detach()).abs().max()) / scale
    assert err <= 5 * floor + 2e-5, (err, floor)
    got = B.keras_weights(base)
    bad, total = [], 0
    for k, p in ctx.params.items():
        if p.grad is None:
            continue
        total += 1
        sc = max(float(p.grad.abs().max()), 1e-300)
        e = float((got[k].grad.detach().cpu().double() - p.grad).abs().max()) / sc
        f = float((ctx32.params[k].grad.double() - p.grad).abs().max()) / sc
        assert e < 0.3, f"{name}: grad {k} rel err {e:.2e}"
        if e >= 5 * f + 1e-4:
            bad.append(f"{k}: {e:.2e} (floor {f:.2e})")
    assert len(bad) <= max(total // 40, 1), f"{name} x {amplitude}: {len(bad)} of {total} tensors off: {bad}"


@pytest.mark.parametrize("name", ["simple", "simple2"])
def test_captured_step_replays_the_range_passes(dev, name):
    """The passes that leave a range zero their slot with a kernel of their own: a hipMemsetAsync node did not zero it when the
    captured step was replayed (every replayed step then saw the maximum of all steps before it — or garbage — and `simple`
    collapsed onto the fallback triplet).  Eager and captured steps give the same losses and triplet counts, bit for bit."""
    from embeddingnet_amd.backbones import get_backbone
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import TripletTrainer

    def run(graph, steps=16):
        torch.manual_seed(0)
        base, _ = get_backbone(SHAPES[name], encodings_len=128, backbone_name=name, backbone_weights=None, seed=3, device=dev)
        base.train()
        tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "radam", 1e-3), k_classes=8, k_samples=4, margin=0.5,
                            negatives_selection_mode="semihard", graph=graph)
        g = torch.Generator().manual_seed(9)
        out = []
        for _ in range(steps):
            x = torch.rand((32,) + SHAPES[name], generator=g).to(dev)
            out.append((float(tr.step(x)), int(tr.last_triplets[1][0].item())))
        assert not tr.ctx.leftovers(), tr.ctx.leftovers()
        return out, tr

    eager, _ = run(False)
    replay, tr = run(True)
    assert tr._graph is not None, getattr(tr, "_graph_error", None)
    assert eager == replay, (eager, replay)
    assert min(c for _, c in replay) > 1                     # (never the fallback triplet)
