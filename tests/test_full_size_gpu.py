"""GPU tests at the sizes BASELINE.json's configs are benchmarked at (C2, C3, C5) — the tail-split planner,
the 150-slab weight-gradient reduce and the 2 GiB operand guard only engage at these sizes.

  * batch-slice consistency: for every distinct convolution geometry of ResNet18 / ResNet50 / EfficientNet-B0 at
    224x224, forward and data gradient of the full-batch launch equal the small-batch launches slice by slice, and the
    full-batch weight gradient equals the sum of the slices' weight gradients; slice 0 of every ResNet18 geometry is
    itself checked against the float64 oracle convolution;
  * C3: SiameseNet(resnet50, 'l2') + contrastive_loss, one 2-branch step vs the float64 oracle at 128x128 x 8 pairs,
    and a 256-pair 224x224 run (finite, repeatable, loss goes down);
  * C5 (one rank's share): EfficientNet-B0 + 'semihard' through TripletTrainer vs the oracle composition with the
    reference's candidate-set rule, and a 64x4 = 256-image 224x224, E = 512 run.
Tolerances are fp32 summation-order bounds, stated where used.
"""
import numpy as np
import pytest
import torch

from oracle import backbones as OB
from oracle import losses as olosses
from oracle import mining as omining

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def g(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def rel(got, want):
    return (got.double() - want.double()).abs().max().item() / max(want.double().abs().max().item(), 1e-30)


def _geometries(name, dev, image=224):
    """Distinct (kind, h, w, cin, k, cout, stride, padding) of the net's conv layers at `image`, from a 1-image trace."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import layers as L
    base, _ = B.get_backbone((image, image, 3), encodings_len=64, backbone_name=name, backbone_weights=None, device=dev)
    seen, hooks = [], []

    def rec(mod, inp):
        x = inp[0].raw if isinstance(inp[0], L.Deferred) else inp[0]
        if isinstance(mod, L.DepthwiseConv2D):
            key = ("dw", x.shape[1], x.shape[2], x.shape[3], mod.k, x.shape[3], mod.stride, "same")
        else:
            key = ("conv", x.shape[1], x.shape[2], x.shape[3], mod.k, mod.kernel.shape[3], mod.stride, mod.padding)
        if key not in seen:
            seen.append(key)

    for m in base.modules():
        if isinstance(m, (L.Conv2D, L.DepthwiseConv2D)):
            hooks.append(m.register_forward_pre_hook(rec))
    base.eval()                                   # inference path: every conv is called as a module (no fused stem)
    fused_pair, L.conv_pair = L.conv_pair, lambda x, c1, c2, emit_stats=False: (c1(x), c2(x))   # ditto for conv pairs
    try:
        with torch.no_grad():
            base(torch.rand((1, image, image, 3), device=dev))
    finally:
        L.conv_pair = fused_pair
        for h in hooks:
            h.remove()
    del base
    torch.cuda.empty_cache()
    return seen


def _layer(key, dev, seed):
    from embeddingnet_amd import layers as L
    kind, h, w, cin, k, cout, stride, padding = key
    gen = torch.Generator().manual_seed(seed)
    if kind == "dw":
        return L.DepthwiseConv2D(cin, k, stride, gen=gen).to(dev)
    return L.Conv2D(cin, cout, k, strides=stride, padding=padding, use_bias=False, kernel_initializer="he_uniform",
                    gen=gen).to(dev)


def _run(layer, x, dy=None):
    xt = x.clone().requires_grad_(True)
    y = layer(xt)
    if dy is None:
        return y.detach()
    w = layer.kernel if hasattr(layer, "kernel") else layer.depthwise_kernel
    w.grad = None
    y.backward(dy)
    return y.detach(), xt.grad, w.grad.clone()


def _slice_consistency(key, dev, n_full, n_piece, seed, oracle_slice0=False):
    kind, h, w, cin, k, cout, stride, padding = key
    layer = _layer(key, dev, seed)
    gen = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((n_full, h, w, cin), device=dev, generator=gen)
    y = _run(layer, x[:1])
    dy = torch.randn((n_full,) + tuple(y.shape[1:]), device=dev, generator=gen)
    yf, dxf, dwf = _run(layer, x, dy)
    dw_sum = torch.zeros_like(dwf, dtype=torch.float64)
    for s in range(0, n_full, n_piece):
        ys, dxs, dws = _run(layer, x[s:s + n_piece].contiguous(), dy[s:s + n_piece].contiguous())
        # same operands in the same K order except where a K-split tail tile sums its parts in another order
        # (fp32 chains of up to R*S*C terms regrouped: a few ulp of the largest output)
        assert rel(yf[s:s + n_piece], ys) < 5e-6, (key, "fwd", s)
        assert rel(dxf[s:s + n_piece], dxs) < 5e-6, (key, "dgrad", s)
        dw_sum += dws.double()
        if s == 0 and oracle_slice0:
            wkey = "c/kernel"
            wt = (layer.kernel if kind == "conv" else layer.depthwise_kernel).detach().cpu().double().requires_grad_(True)
            ctx = OB.Ctx({wkey: wt} if kind == "conv" else {"c/depthwise_kernel": wt})
            xr = x[:n_piece].cpu().double().requires_grad_(True)
            if kind == "conv":
                yr = OB.conv2d(ctx, "c", xr, cout, k, stride=stride, padding=padding, bias=False)
            else:
                yr = OB.depthwise_conv2d(ctx, "c", xr, k, stride, OB.conv_normal)
            yr.backward(dy[:n_piece].cpu().double())
            assert rel(ys.cpu(), yr.detach()) < 2e-5, (key, "fwd vs oracle")
            assert rel(dxs.cpu(), xr.grad) < 2e-5, (key, "dgrad vs oracle")
            assert rel(dws.cpu(), wt.grad) < 2e-5 * max(1.0, (n_piece * h * w / stride ** 2 / 2048) ** 0.5), (key, "wgrad vs oracle")
    # full-batch split-K slabs vs the pieces' own split-K sums: fp32 chains of length N*OH*OW in different groupings
    assert rel(dwf, dw_sum) < 2e-5, (key, "wgrad", rel(dwf, dw_sum))


def test_resnet18_full_batch_conv_geometries(dev):
    """C2: every conv geometry of ResNet18 @224 at the benchmark's local batch 128 vs 16 launches of 8 images; slice 0
    of each against the float64 oracle.  (The stem runs on the 4-channel padded image in training; both forms here.)"""
    geoms = _geometries("resnet18", dev)
    assert len(geoms) >= 11, geoms
    geoms.append(("conv", 224, 224, 4, 7, 64, 2, 3))              # the fused stem's padded-channel conv0
    for i, key in enumerate(geoms):
        _slice_consistency(key, dev, 128, 8, seed=100 + i, oracle_slice0=True)
        torch.cuda.empty_cache()


# ---- the DEFAULT arithmetic at full size -------------------------------------------------------------------------
# The tests above build bare Conv2D layers: six-term gather kernels.  The zoo ResNets run something else (backbones._rn_conv: f16 = True):
# 3x3 stride-1 convs on the patch kernel / the planes weight gradient, every other conv on the three-product gather kernels — all of
# them fed operand ranges / planes by the BatchNormalizations around them.  Here every geometry runs THAT way at the benchmark's
# batch, with the hand-overs a BatchNormalization would make (input planes + range slot on the tensor, gradient planes / range slot
# in the step context), slice by slice against the small-batch launches and slice 0 against the float64 oracle; the kernel trace
# must hold no six-term conv kernel.
def _range_slot(t):
    from embeddingnet_amd import _lib
    lib = _lib.lib()
    slot = torch.zeros(lib.embnet_range_slot_words(), dtype=torch.int32, device=t.device)
    table = torch.tensor([[t.data_ptr(), t.numel(), slot.data_ptr()]], dtype=torch.int64, device=t.device)
    ce = lib.embnet_range_chunk_elems()
    chunks = torch.tensor([(0, j) for j in range(-(-t.numel() // ce))], dtype=torch.int32, device=t.device)
    _lib.check(lib.embnet_range_multi(table.data_ptr(), 1, chunks.data_ptr(), chunks.shape[0], _lib.stream()))
    return slot


def _planes(t):
    from embeddingnet_amd import _lib
    c = t.shape[-1]
    p = torch.empty(3 * t.numel(), device=t.device, dtype=torch.int16)
    _lib.check(_lib.lib().embnet_planes_from_f32(t.data_ptr(), t.numel() // c, c, p.data_ptr(), _lib.stream()))
    return p


def _run_default(layer, x, dy):
    """forward + backward of a zoo-ResNet conv the way the net runs it: x carries its range (and planes, for a patch conv), dy's planes /
    range wait in the step context under its address, as the BatchNormalization behind the conv would have left them."""
    from embeddingnet_amd import layers as L
    xt = x.clone().requires_grad_(True)
    xt._range = _range_slot(xt)
    if layer.patch_capable(xt.shape):
        xt._planes = _planes(xt)
    y = layer(xt)
    layer.kernel.grad = None
    dy = dy.contiguous()
    if getattr(y, "_wants_dy_planes", False):
        L.DY_PLANES[dy.data_ptr()] = (_planes(dy), dy)
    if getattr(y, "_wants_dy_range", False):
        L.DY_RANGE[dy.data_ptr()] = (_range_slot(dy), dy)
    y.backward(dy)
    assert not L.current_context().leftovers(), L.current_context().leftovers()
    return y.detach(), xt.grad, layer.kernel.grad.clone()


SIX_TERM = ("void embnet::conv_fwd_kernel<", "void embnet::conv_dgrad_kernel<", "void embnet::conv_wgrad_kernel<",
            "void embnet::conv_fwd_tf_kernel<", "void embnet::conv_wgrad_tf_kernel<")


def _slice_consistency_default(key, dev, n_full, n_piece, seed, oracle_slice0=True):
    from embeddingnet_amd import _lib
    kind, h, w, cin, k, cout, stride, padding = key
    layer = _layer(key, dev, seed)
    layer.f16 = True
    gen = torch.Generator(device=dev).manual_seed(seed)
    x = torch.relu(torch.randn((n_full, h, w, cin), device=dev, generator=gen)) * 0.05      # behind BatchNorm + ReLU, small gamma
    y = _run(layer, x[:1])
    dy = torch.randn((n_full,) + tuple(y.shape[1:]), device=dev, generator=gen) * 1e-5
    _lib.trace_reset(); _lib.trace_enable(True)
    try:
        yf, dxf, dwf = _run_default(layer, x, dy)
        names = [r[0] for r in _lib.trace_records()]
    finally:
        _lib.trace_enable(False)
    assert not any(s.startswith(SIX_TERM) for s in names), (key, [s for s in names if s.startswith(SIX_TERM)])
    assert any(("_h_kernel" in s or "conv_patch_kernel" in s) for s in names), (key, names)
    dw_sum = torch.zeros_like(dwf, dtype=torch.float64)
    for s in range(0, n_full, n_piece):
        ys, dxs, dws = _run_default(layer, x[s:s + n_piece].contiguous(), dy[s:s + n_piece].contiguous())
        # (the slices' operands carry their own ranges: another power-of-two scale, the same 22 bits — plus the K-split regrouping above)
        assert rel(yf[s:s + n_piece], ys) < 5e-6, (key, "fwd", s)
        assert rel(dxf[s:s + n_piece], dxs) < 5e-6, (key, "dgrad", s)
        dw_sum += dws.double()
        if s == 0 and oracle_slice0:
            wt = layer.kernel.detach().cpu().double().requires_grad_(True)
            ctx = OB.Ctx({"c/kernel": wt})
            xr = x[:n_piece].cpu().double().requires_grad_(True)
            yr = OB.conv2d(ctx, "c", xr, cout, k, stride=stride, padding=padding, bias=False)
            yr.backward(dy[:n_piece].cpu().double())
            assert rel(ys.cpu(), yr.detach()) < 2e-5, (key, "fwd vs oracle")
            assert rel(dxs.cpu(), xr.grad) < 2e-5, (key, "dgrad vs oracle")
            assert rel(dws.cpu(), wt.grad) < 2e-5 * max(1.0, (n_piece * h * w / stride ** 2 / 2048) ** 0.5), (key, "wgrad vs oracle")
    assert rel(dwf, dw_sum) < 2e-5, (key, "wgrad", rel(dwf, dw_sum))


def test_resnet18_full_batch_default_arithmetic(dev):
    """C2: every conv geometry of ResNet18 @224 at batch 128 on the kernels the net runs (three products: patch / planes /
    ranged gather kernels) vs 16 launches of 8 images; slice 0 of each against the float64 oracle."""
    geoms = [g_ for g_ in _geometries("resnet18", dev) if g_[3] % 4 == 0]
    geoms.append(("conv", 224, 224, 4, 7, 64, 2, 3))              # the fused stem's padded-channel conv0
    assert len(geoms) >= 11, geoms
    for i, key in enumerate(geoms):
        _slice_consistency_default(key, dev, 128, 8, seed=400 + i)
        torch.cuda.empty_cache()


def test_resnet50_full_batch_default_arithmetic(dev):
    """C3's backbone: every conv geometry of ResNet50 @224 at batch 256 (one Siamese branch) on the kernels the net runs, vs 16
    launches of 16; slice 0 against the float64 oracle."""
    geoms = [g_ for g_ in _geometries("resnet50", dev) if g_[3] % 4 == 0]
    assert len(geoms) >= 19, geoms
    for i, key in enumerate(geoms):
        _slice_consistency_default(key, dev, 256, 16, seed=500 + i)
        torch.cuda.empty_cache()


def test_c2_full_size_training_step(dev):
    """One C2 step at full size (ResNet18, 224x224, 32 x 4 = 128 images, 'hardest') through TripletTrainer: finite, bit for bit
    repeatable from the same start, and its kernel trace holds no six-term conv kernel — the default arithmetic end to end."""
    from embeddingnet_amd import _lib
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import TripletTrainer

    def run(trace):
        base, _ = B.get_backbone((224, 224, 3), encodings_len=256, backbone_name="resnet18", backbone_weights=None, seed=3, device=dev)
        base.train()
        tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "radam", 1e-4), 32, 4, margin=0.5, negatives_selection_mode="hardest",
                            graph=False, seed=1)
        gen = torch.Generator(device=dev).manual_seed(7)
        losses, names = [], []
        for i in range(3):
            x = torch.rand((128, 224, 224, 3), device=dev, generator=gen)
            if trace and i == 1:
                _lib.trace_reset(); _lib.trace_enable(True)
            losses.append(float(tr.step(x)))
            if trace and i == 1:
                names = [r[0] for r in _lib.trace_records()]
                _lib.trace_enable(False)
        left = tr.ctx.leftovers()
        unclaimed = dict(tr.ctx.unclaimed)
        del tr, base
        torch.cuda.empty_cache()
        return losses, names, left, unclaimed

    l1, names, left, unclaimed = run(True)
    l2, _, _, _ = run(False)
    assert all(np.isfinite(l1)) and l1 == l2, (l1, l2)
    assert not left, left
    print("unclaimed hand-overs over three steps:", unclaimed)
    six = [s for s in names if s.startswith(SIX_TERM)]
    assert not six, six
    assert sum("conv_patch_kernel" in s for s in names) >= 26 and sum("conv_wgrad_planes_kernel" in s for s in names) >= 13, names
    assert sum("conv_stem_kernel" in s for s in names) == 1                     # the 7x7 / 2 stem's forward: csrc/conv_stem.hip
    assert sum("_h_kernel" in s for s in names) >= 7 + 7 + 8, [s for s in names if "_h_kernel" in s]
    # every gradient written as planes takes its scale from a bound: no dry run of a BatchNorm-backward apply pass
    assert not any("bn_bwd_apply4_kernel<1>" in s for s in names) and sum("bn_bwd_apply4_kernel<4>" in s for s in names) >= 13, \
        [s for s in names if "bn_bwd_apply4" in s]


def test_resnet50_full_batch_conv_geometries(dev):
    """C3's backbone: every conv geometry of ResNet50 @224 at batch 256 (one Siamese branch) vs 16 launches of 16."""
    geoms = _geometries("resnet50", dev)
    assert len(geoms) >= 20, geoms
    for i, key in enumerate(geoms):
        _slice_consistency(key, dev, 256, 16, seed=200 + i)
        torch.cuda.empty_cache()


def test_efficientnet_b0_full_batch_conv_geometries(dev):
    """C5's backbone: every conv and depthwise geometry of EfficientNet-B0 @224 at local batch 256 vs 16 launches of 16."""
    geoms = _geometries("efficientnet-b0", dev)
    assert sum(k[0] == "dw" for k in geoms) >= 8 and len(geoms) >= 30, geoms
    for i, key in enumerate(geoms):
        _slice_consistency(key, dev, 256, 16, seed=300 + i)
        torch.cuda.empty_cache()


def _siamese(name, image, enc, dev, distance_type="l2", seed=4):
    from embeddingnet_amd.models import SiameseNet
    params = {"model": dict(input_shape=[image, image, 3], encodings_len=enc, mode="siamese", distance_type=distance_type,
                            backbone_name=name, backbone_weights=None, freeze_backbone=False,
                            embeddings_normalization=True, device=dev, seed=seed),
              "dataloader": {}, "generator": {}, "train": {}, "general": {"work_dir": "w/", "project_name": "p"}}
    net = SiameseNet(params, training=True)
    for m in net.model.modules():
        if hasattr(m, "enabled"):
            m.enabled = False
    return net


def test_c3_siamese_resnet50_step_vs_oracle(dev):
    """SiameseNet(resnet50, 'l2') -> [distance, cls1, cls2]; contrastive_loss on the distance: loss within 1e-4 relative
    of the float64 oracle, gradients within fp32 bounds; the two classification outputs are the sigmoid head on each
    embedding (models.py:44, 211-215)."""
    from embeddingnet_amd.backbones import keras_weights
    from embeddingnet_amd.losses_and_accuracies import accuracy, contrastive_loss
    image, enc, b = 128, 32, 8
    net = _siamese("resnet50", image, enc, dev)
    rs = np.random.RandomState(2)
    x1, x2 = rs.rand(b, image, image, 3).astype(np.float32), rs.rand(b, image, image, 3).astype(np.float32)
    x2[: b // 2] = np.clip(x1[: b // 2] + 0.05 * rs.randn(b // 2, image, image, 3), 0, 1)     # "same class" pairs
    y = np.zeros((b, 1), np.float32)
    y[: b // 2] = 1
    net.model.train()
    out = net.model([g(x1, dev), g(x2, dev)])
    assert len(out) == 3 and all(tuple(o.shape) == (b, 1) for o in out)
    loss = contrastive_loss(g(y, dev), out[0])
    acc = accuracy(g(y, dev), out[0])
    loss.backward()
    W = {k: v.detach().cpu().double().requires_grad_(v.requires_grad) for k, v in keras_weights(net.model).items()}
    ctx = OB.Ctx(W, training=True)
    kw = dict(backbone_name="resnet50", encodings_len=enc)
    e1 = OB.base_model(ctx, torch.tensor(x1, dtype=torch.float64), **kw)
    e2 = OB.base_model(ctx, torch.tensor(x2, dtype=torch.float64), **kw)
    d = OB.siamese_l2_distance(e1, e2)
    yt = torch.tensor(y, dtype=torch.float64)
    want = (yt * d ** 2 + (1 - yt) * torch.clamp(1 - d, min=0) ** 2).mean()
    assert abs(want.item() - olosses.contrastive_loss(y, d.detach().numpy())) < 1e-12
    assert abs(loss.item() - want.item()) <= 1e-4 * abs(want.item()), (loss.item(), want.item())
    assert acc.item() == pytest.approx(olosses.accuracy(y, d.detach().numpy()))
    for o, e in ((out[1], e1), (out[2], e2)):
        assert rel(o.detach().cpu(), OB.classification_head(ctx, e).detach()) < 1e-4
    want.backward()
    got = keras_weights(net.model)
    num = den = 0.0
    for k, p in W.items():
        if p.grad is None:
            continue
        diff = got[k].grad.detach().cpu().double() - p.grad
        num += (diff ** 2).sum().item(); den += (p.grad ** 2).sum().item()
        assert diff.abs().max().item() / max(p.grad.abs().max().item(), 1e-12) < 0.3, k
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5     # global bound as test_backbone_forward_backward_vs_oracle


def test_c3_full_size_256_pairs(dev):
    """C3 at its benchmarked size: ResNet50 224x224, 256 pairs (first half same class, datagenerators.py:345-374), 'l2'
    head + contrastive_loss, Adam: finite, bit-repeatable from the same state, and the loss goes down."""
    from embeddingnet_amd.losses_and_accuracies import contrastive_loss
    from embeddingnet_amd.optimizers import KerasOptimizer
    pairs = 256
    net = _siamese("resnet50", 224, 256, dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    proto = torch.rand((16, 224, 224, 3), device=dev, generator=gen)
    cls = torch.randint(0, 16, (pairs,), device=dev, generator=gen)
    other = (cls + torch.randint(1, 16, (pairs,), device=dev, generator=gen)) % 16
    other[: pairs // 2] = cls[: pairs // 2]
    noise = lambda: 0.1 * torch.randn((pairs, 224, 224, 3), device=dev, generator=gen)
    x1, x2 = (proto[cls] + noise()).clamp_(0, 1), (proto[other] + noise()).clamp_(0, 1)
    y = torch.zeros((pairs, 1), device=dev)
    y[: pairs // 2] = 1
    opt = KerasOptimizer([p for p in net.model.parameters() if p.requires_grad], "adam", 1e-4)
    net.model.train()

    def loss_only():
        with torch.no_grad():
            return contrastive_loss(y, net.model([x1, x2])[0]).item()

    a, b = loss_only(), loss_only()
    assert np.isfinite(a) and a == b, (a, b)                     # deterministic kernels: same state, same bits
    hist = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = contrastive_loss(y, net.model([x1, x2])[0])
        loss.backward()
        opt.step()
        hist.append(loss.item())
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist
    for p in net.model.parameters():
        assert torch.isfinite(p).all()


def _semihard_sets_ok(trip, emb_oracle, p, k, margin, tol=2e-5):
    """Every device-mined (a,p,n) has n in the reference's semihard candidate set of that pair (computed on the
    oracle's embeddings), pairs without candidates yield no triplet; borderline loss values (within tol of 0 or of the
    margin) may fall either way."""
    n = p * k
    d = omining.pairwise_distances(emb_oracle.astype(np.float32))
    gpu = {(int(a), int(b)): int(c) for a, b, c in trip}
    for c in range(p):
        lo, hi = c * k, (c + 1) * k
        neg = np.concatenate([np.arange(0, lo), np.arange(hi, n)])
        for i in range(lo, hi):
            for j in range(i + 1, hi):
                lv = d[i, j] - d[i, neg] + margin
                strict = (lv > tol) & (lv < margin - tol)
                loose = (lv > -tol) & (lv < margin + tol)
                got = gpu.get((i, j))
                if got is None:
                    assert not strict.any(), (i, j, "no triplet although the pair has semihard candidates")
                else:
                    assert loose[int(np.where(neg == got)[0][0])], (i, j, got)


def test_c5_local_efficientnet_semihard_step_vs_oracle(dev):
    """One rank's C5 step in miniature: EfficientNet-B0 + 'semihard' through the fused TripletTrainer.  Mined negatives
    must come from the reference's candidate sets; the loss on the device's triplets within 1e-4 relative of the
    float64 oracle composition; gradients within the fp32 bounds used for whole backbones."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.train_step import TripletTrainer
    name, shape, enc, p, k, m = "efficientnet-b0", (64, 64, 3), 64, 8, 4, 0.5
    base, _ = B.get_backbone(shape, encodings_len=enc, backbone_name=name, backbone_weights=None, seed=2, device=dev)
    for mod in base.modules():
        if hasattr(mod, "enabled"):
            mod.enabled = False
    rs = np.random.RandomState(1)
    cls = rs.rand(p, *shape)
    x = np.clip(np.repeat(cls, k, axis=0) + 0.25 * rs.randn(p * k, *shape), 0, 1).astype(np.float32)
    tr = TripletTrainer(base, None, p, k, margin=m, negatives_selection_mode="semihard", seed=3)
    base.train()
    total, mean, count = tr.loss(g(x, dev))
    total.backward()
    trip = tr.last_triplets[0][: int(count.item())].cpu().numpy()
    W = {kk: v.detach().cpu().double().requires_grad_(v.requires_grad) for kk, v in B.keras_weights(base).items()}
    ctx = OB.Ctx(W, training=True)
    emb = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name=name, encodings_len=enc)
    _semihard_sets_ok(trip, emb.detach().numpy(), p, k, m)
    t = torch.as_tensor(trip, dtype=torch.long)
    rows = torch.clamp(((emb[t[:, 0]] - emb[t[:, 1]]) ** 2).sum(1) - ((emb[t[:, 0]] - emb[t[:, 2]]) ** 2).sum(1) + m, min=0)
    assert len(trip) >= 1
    assert abs(mean.item() - rows.mean().item()) <= 1e-4 * abs(rows.mean().item()), (mean.item(), rows.mean().item())
    rows.mean().backward()
    got = B.keras_weights(base)
    num = den = 0.0
    for kk, pr in W.items():
        if pr.grad is None:
            continue
        diff = got[kk].grad.detach().cpu().double() - pr.grad
        num += (diff ** 2).sum().item(); den += (pr.grad ** 2).sum().item()
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_c5_full_size_local_step(dev):
    """C5's per-rank workload: EfficientNet-B0 224x224, 64 classes x 4 = 256 images, E = 512, semihard, RAdam."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import TripletTrainer
    p, k = 64, 4
    base, _ = B.get_backbone((224, 224, 3), encodings_len=512, backbone_name="efficientnet-b0", backbone_weights=None,
                             seed=0, device=dev)
    gen = torch.Generator(device=dev).manual_seed(6)
    proto = torch.rand((p, 224, 224, 3), device=dev, generator=gen)
    x = (proto.repeat_interleave(k, 0) + 0.2 * torch.randn((p * k, 224, 224, 3), device=dev, generator=gen)).clamp_(0, 1)
    opt = KerasOptimizer([q for q in base.parameters() if q.requires_grad], "radam", 1e-3)
    tr = TripletTrainer(base, opt, p, k, margin=0.5, negatives_selection_mode="semihard", seed=1)
    hist = [tr.step(x).item() for _ in range(8)]
    assert all(np.isfinite(hist)), hist
    trip, count = tr.last_triplets
    c = int(count.item())
    assert 1 <= c <= p * k * (k - 1) // 2
    t = trip[:c].cpu().numpy()
    assert (t[:, 0] // k == t[:, 1] // k).all() and (t[:, 0] // k != t[:, 2] // k).all() and (t[:, 0] < t[:, 1]).all()
    for q in base.parameters():
        assert torch.isfinite(q).all()
