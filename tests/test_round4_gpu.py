"""Round-4 GPU tests: the state a bench must never time (the reference's fallback triplet with hinge 0), bench.py's
liveness record and its self-spawned ranks, data-parallel EfficientNet (odd-sized parameters in the flat gradient buffer).

Oracle = test infrastructure (oracle/*.py); every device result goes through the C ABI of libembnet_hip.so.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib
from oracle import mining as omining
from oracle import optimizers as OO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------ the fallback step
@pytest.mark.parametrize("rule", ["adam", "radam", "sgd"])
def test_fallback_step_has_exactly_zero_gradients(dev, rule):
    """/root/reference/embedding_net/datagenerators.py:246-250: when mining selects nothing, ONE triplet is emitted (last
    pair of the last class + the first negative).  If that triplet's hinge is 0 the step's loss is 0, every gradient is
    EXACTLY zero, and the weights move only by what the optimizer's moment slots still hold (oracle: the NumPy rule fed
    zero gradients).  Forced here with a margin no distance on the unit sphere can violate."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import TripletTrainer
    p, k, lr = 4, 3, 1e-3
    base, _ = B.get_backbone((64, 64, 3), encodings_len=32, backbone_name="resnet18", backbone_weights=None, seed=11, device=dev)
    params = [q for q in base.parameters() if q.requires_grad]
    opt = KerasOptimizer(params, rule, lr)
    tr = TripletTrainer(base, opt, p, k, margin=0.5, negatives_selection_mode="hardest")
    gen = torch.Generator(device=dev).manual_seed(21)
    oo = OO.get_optimizer(rule, lr)
    for _ in range(7):                                  # live steps: the moment slots fill up (RAdam passes its warm-up)
        loss = tr.step(torch.rand((p * k, 64, 64, 3), device=dev, generator=gen))
        assert loss.item() > 0 and int(tr.last_triplets[1].item()) > 1
        oo.t += 1
    # hand the oracle the device's slots (float64 copies), then both take the dead step
    names = {"adam": ("m", "v"), "radam": ("m", "v"), "sgd": ()}[rule]
    for i, q in enumerate(params):
        st = opt.state.get(q, {})
        for slot, nm in zip(("slot1", "slot2"), names):
            oo.slots[(i, nm)] = st[slot].detach().cpu().double().numpy().copy()
    before = [q.detach().cpu().double().numpy().copy() for q in params]
    tr.margin = -10.0
    x = torch.rand((p * k, 64, 64, 3), device=dev, generator=gen)
    loss = tr.step(x)
    trip, count = tr.last_triplets
    n = p * k
    assert int(count.item()) == 1 and trip[0].tolist() == [n - 2, n - 1, 0], (count, trip[0])
    want = omining.mine_triplets(np.ones((n, n), np.float32) - np.eye(n, dtype=np.float32), p, k, -10.0, "hardest")
    assert want["fallback"] and want["triplets"].tolist() == [trip[0].tolist()]
    assert loss.item() == 0.0 and tr.last_total.item() == 0.0
    for q in params:
        assert q.grad is not None and float(q.grad.abs().max()) == 0.0, "a dead step must have exactly zero gradients"
    ref = [w.copy() for w in before]
    oo.step(ref, [np.zeros_like(w) for w in ref])
    moved = 0.0
    for q, w0, w1 in zip(params, before, ref):
        got = q.detach().cpu().double().numpy()
        moved = max(moved, float(np.abs(w1 - w0).max()))
        assert np.abs(got - w1).max() <= 3e-6 * max(np.abs(w1).max(), 1e-30) + 1e-12
    assert (moved == 0.0) == (rule == "sgd"), "momentum rules keep moving on their slots, plain SGD stands still"


# ------------------------------------------------------------------------------------------------ bench.py
def _bench(*extra, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                          "--sustain-seconds", "0", *extra], capture_output=True, text=True,
                         timeout=timeout, env=dict(os.environ, **(env or {})))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def test_bench_reports_a_live_problem_and_refuses_a_dead_one():
    """The timed region must run on live gradients: bench.py cycles resident batches, records every timed step's loss and
    mined-triplet count on the device, prints them, and exits non-zero without a JSON line if any timed step had loss 0 or
    only the fallback triplet (forced here by an unviolable margin)."""
    out, d = _bench("--config", "c1", "--pool", "4")
    assert out.returncode == 0 and d is not None, out.stderr[-3000:]
    c = d["config"]
    assert c["resident_batches"] == 4 and c["loss_first_timed"] > 0 and c["loss_min_timed"] > 0 and c["active_triplets_min"] > 1
    assert "4 resident batches" in c["workload"]
    out, d = _bench("--config", "c1", "--margin", "-10")
    assert out.returncode != 0 and d is None and "dead problem" in out.stderr, out.stderr[-2000:]
    out, d = _bench("--config", "c1", "--margin", "-10", "--allow-dead")
    assert out.returncode == 0 and d["config"]["loss_min_timed"] == 0.0 and d["config"]["active_triplets_min"] == 1


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher on the command line, the form of the driver's N = 1 command): the parent
    spawns one child per rank BEFORE touching the GPU and relays rank 0's JSON line.  Two ranks share this box's one GPU
    over gloo (RCCL refuses two ranks per device)."""
    out, d = _bench("--gpus", "2", "--backbone", "resnet18", "--image", "64", "--k-classes", "8",
                    env={"EMBNET_DIST_BACKEND": "gloo"})
    assert out.returncode == 0 and d is not None, out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * 8 * 4 and d["config"]["parallelism"] == "dp2"
    assert len(d["config"]["host_enqueue_ms_per_step_by_rank"]) == 2 and d["config"]["loss_min_timed"] > 0


def test_bench_world_of_eight_on_one_gpu():
    """The N = 8 control flow on hardware that has one GPU: eight ranks over gloo share it, tiny shapes (simple2 64x64, two
    classes per rank).  What has then run at world size 8: the rendezvous, the model broadcast, the first-backward bucket order
    agreement, the collective capture decision (TripletTrainer._agree), eight bucketed gradient all-reduces per step, the MAX
    of the elapsed times, the liveness MAX, the per-rank record in rank 0's line.  (RCCL itself: the driver's scaling run.)"""
    out, d = _bench("--gpus", "8", "--config", "c1", "--k-classes", "2", "--pool", "8", "--steps", "6", "--warmup", "4",
                    "--mining", "hardest", "--allow-dead",       # (8 images per rank: a step without a violating triplet is possible)
                    env={"EMBNET_DIST_BACKEND": "gloo"})
    assert out.returncode == 0 and d is not None, out.stderr[-3000:]
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * 2 * 4 and d["config"]["parallelism"] == "dp8"
    ranks = d["config"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and len({r["step_mode"] for r in ranks}) == 1
    assert all(r["gradient_mean"] == "sum + scale" for r in ranks) and d["config"]["loss_first_timed"] > 0
    assert out.stderr.count("gradient mean:") == 8


# ------------------------------------------------------------------------------------------------ DP with odd-sized slots
def test_dp_efficientnet_gradients_in_place(dev):
    """EfficientNet-B0 has parameters whose size is not a multiple of 4 (squeeze-excite reduce biases of 6 / 10 elements):
    the flat gradient buffer keeps every slot on a 16-byte boundary, so the kernels that write gradients in place (16-byte
    stores) accept every slot, and the in-place gradients equal plain autograd's."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.parallel import GradReducer
    from embeddingnet_amd.train_step import TripletTrainer
    x = torch.rand((12, 64, 64, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    flats = []
    for direct in (False, True):
        base, _ = B.get_backbone((64, 64, 3), encodings_len=32, backbone_name="efficientnet-b0", backbone_weights=None, seed=2, device=dev)
        for m in base.modules():
            if hasattr(m, "enabled"):
                m.enabled = False                               # dropout / drop-connect off: both runs see the same graph
        params = [p for p in base.parameters() if p.requires_grad]
        assert any(p.numel() % 4 for p in params)
        opt = KerasOptimizer(params, "sgd", 0.0)
        red = GradReducer(params)
        tr = TripletTrainer(base, opt, 4, 3, margin=0.5, negatives_selection_mode="hardest", reducer=red)
        red.direct(direct)
        tr.step(x)
        tr.step(x)
        for p in params:
            assert p.grad.data_ptr() % 16 == 0
        flats.append([red.flat[off:off + n].clone() for off, n in (red._slot[p] for p in params)])
        red.close()
    for a, b in zip(*flats):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ deferred slab sums
@pytest.mark.parametrize("backbone,shape", [("resnet18", (64, 64, 3)), ("simple2", (64, 64, 3))])
def test_deferred_slab_sums_are_bit_identical(dev, backbone, shape):
    """The weight gradients' split-K slab sums queued during backward and added up by ONE launch
    (embnet_slab_reduce_multi) equal the per-layer sums bit for bit: same loss, gradients and updated weights."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import layers as L
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import TripletTrainer
    x = torch.rand((12,) + shape, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    res, launches = [], []
    for defer in (False, True):
        L.SLAB_DEFER_ENABLED[0] = defer
        try:
            base, _ = B.get_backbone(shape, encodings_len=32, backbone_name=backbone, backbone_weights=None, seed=4, device=dev)
            for m in base.modules():
                if hasattr(m, "enabled"):
                    m.enabled = False
            params = [p for p in base.parameters() if p.requires_grad]
            tr = TripletTrainer(base, KerasOptimizer(params, "adam", 1e-3), 4, 3, margin=0.5, negatives_selection_mode="hardest")
            tr.step(x)
            _lib.trace_reset(); _lib.trace_enable(True)
            loss = tr.step(x)
            torch.cuda.synchronize()
            names = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
        finally:
            L.SLAB_DEFER_ENABLED[0] = True
        launches.append((sum("slab_reduce_kernel" in n for n in names), sum("slab_reduce_multi" in n for n in names)))
        res.append((loss.clone(), [p.grad.clone() for p in params], [p.detach().clone() for p in params]))
    # (the ResNet stem's padded-kernel gradient is consumed at once and keeps its own slab sum)
    assert launches[0][0] >= 3 and launches[0][1] == 0 and launches[1][0] <= 1 and launches[1][1] == 1, launches
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ conv -> ReLU -> BN blocks
@pytest.mark.parametrize("shape,k,cout", [((4, 20, 20, 32), 3, 32), ((3, 17, 15, 16), 5, 64), ((2, 9, 9, 64), 4, 128)])
def test_relu_backward_fused_into_batchnorm_backward(dev, shape, k, cout):
    """conv(+bias, ReLU) -> BatchNormalization (simple2's block, reference backbones.py:44-68): with the ReLU's backward and
    the bias gradient folded into the BN backward pass (embnet_bn_bwd_inrelu) and the BN statistics taken from the conv
    epilogue, outputs and all gradients equal the unfused chain — dx, dW, dgamma, dbeta bit for bit where the arithmetic is
    the same, the bias gradient and the epilogue statistics within fp32 summation-order differences — and the fused path
    launches no relu_bwd_colsum / bn_stats kernel."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_RELU_BN[0] = fuse
        try:
            gen = torch.Generator().manual_seed(3)
            conv = L.Conv2D(shape[-1], cout, k, activation="relu", gen=gen).to(dev)
            with torch.no_grad():
                conv.bias.copy_(torch.linspace(-0.3, 0.3, cout))
            bn = L.BatchNormalization(cout).to(dev).train()
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, cout)); bn.beta.copy_(torch.linspace(-0.2, 0.2, cout))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y = bn(conv(xt, emit_stats=fuse))
            y.backward(torch.sin(y.detach() * 2))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(y=y.detach(), dx=xt.grad, dW=conv.kernel.grad, db=conv.bias.grad, dgamma=bn.gamma.grad, dbeta=bn.beta.grad,
                             mm=bn.moving_mean.clone())
        finally:
            L.FUSE_RELU_BN[0] = True
            L.RELU_DONE.clear()
    assert any("relu_bwd_colsum" in n for n in names[False]) and any("bn_stats" in n for n in names[False])
    assert not any("relu_bwd" in n or "bn_stats" in n for n in names[True]), names[True]
    assert any("bn_bwd_apply_inrelu4" in n for n in names[True])
    for key in res[True]:
        a, b = res[True][key], res[False][key]
        scale = b.abs().max().item() + 1e-30
        assert (a - b).abs().max().item() <= 2e-5 * scale, (key, (a - b).abs().max().item(), scale)
    # the masked gradient itself is exact: dx / dW are computed from identical dz values when the statistics agree; with the
    # epilogue statistics they differ in the last bits only
    assert not L.RELU_DONE


# ------------------------------------------------------------------------------------------------ BN -> Dropout
@pytest.mark.parametrize("image,inrelu", [(64, True), (40, True), (64, False)])
def test_dropout_behind_batchnorm_rides_on_its_kernels(dev, image, inrelu):
    """simple2's bn3 -> drop1 and bn6 -> drop2 (reference backbones.py:52-55,63-66): with the Dropout applied inside the
    BatchNormalization's forward apply and backward passes (embnet_affine_act_dropout, embnet_bn_bwd_inrelu_dropout) the
    backbone's output and every gradient equal the separate-layer chain BIT FOR BIT (same counter-based mask, same
    multiply), over two steps (the mask counter advances the same way), and no dropout kernel is launched for them.
    inrelu=False: the ReLU backward is not fused into the BN backward, so the Dropout backward runs as a pass in front of it."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import layers as L
    x = torch.rand((6, image, image, 3), device=dev)
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_DROPOUT_BN[0] = fuse
        L.FUSE_RELU_BN[0] = inrelu
        try:
            base, _ = B.get_backbone((image, image, 3), encodings_len=32, backbone_name="simple2", backbone_weights=None, seed=5)
            base = base.to(dev).train()
            out = []
            _lib.trace_reset(); _lib.trace_enable(True)
            for step in range(2):
                for p in base.parameters():
                    p.grad = None
                y = base(x)
                y.backward(torch.cos(y.detach() * 3 + step))
                out.append([y.detach().clone()] + [p.grad.clone() for p in base.parameters()])
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = out
        finally:
            L.FUSE_DROPOUT_BN[0] = True
            L.FUSE_RELU_BN[0] = True
            L.RELU_DONE.clear()
    n_sep, n_fused = (sum("dropout_kernel" in n for n in names[f]) for f in (False, True))
    # two steps x (3 Dropout layers forward + backward); fused: only the head's Dropout (behind a Dense) keeps its kernel,
    # plus — without the fused ReLU backward — the two backward passes in front of the BN backward
    assert n_sep == 12 and n_fused == (4 if inrelu else 8), (n_sep, n_fused)
    for sa, sb in zip(res[False], res[True]):
        for a, b in zip(sa, sb):
            assert torch.equal(a, b)
    assert not torch.equal(res[True][0][0], res[True][1][0])          # a new mask each step


# ------------------------------------------------------------------------------------------------ BN backward sums from dgrad
@pytest.mark.parametrize("shape,k,ks,with_skip", [((8, 14, 14, 64), 128, 1, False), ((4, 9, 11, 32), 64, 3, False),
                                                  ((6, 7, 7, 256), 64, 1, True), ((3, 20, 20, 128), 256, 1, True),
                                                  # C3's first bottleneck conv at its full size (256 images of one branch):
                                                  # 6 272 whole tiles + K-split left-over tiles through the fix-up kernel
                                                  ((256, 56, 56, 256), 64, 1, True), ((256, 14, 14, 256), 1024, 1, False)])
def test_batchnorm_backward_sums_come_from_the_conv_data_gradient(dev, shape, k, ks, with_skip):
    """BatchNormalization(+ReLU) -> stride-1 Conv2D on the gather kernels (the zoo ResNets' bn -> relu -> 1x1 conv,
    reference backbones.py:99-104): the conv's data-gradient epilogue emits the BatchNorm-backward column sums
    (embnet_conv2d_dgrad_bnsums_f32), the BatchNormalization backward starts at its finalize kernel
    (embnet_bn_bwd_partials) and no bn_bwd_reduce kernel runs.  dx of the BN, dgamma, dbeta and the conv's gradients equal the
    separate-pass chain within fp32 summation-order differences; with_skip: the BN's input also feeds a skip connection whose
    gradient is folded into the BN backward apply (dx_add)."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_BN_SUMS[0] = fuse
        try:
            gen = torch.Generator().manual_seed(11)
            c = shape[-1]
            bn = L.BatchNormalization(c, relu=True).to(dev).train()
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            conv = L.Conv2D(c, k, ks, padding="same", use_bias=False, gen=gen).to(dev)
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            if with_skip:
                a, skip = bn(xt, with_skip=True)
                y = conv(a)
                out = y.sum(dim=-1, keepdim=True) * 0.01 + skip          # a second path from the BN's input
            else:
                y = conv(bn(xt))
                out = y
            out.backward(torch.cos(out.detach() * 1.7))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(dx=xt.grad, dW=conv.kernel.grad, dgamma=bn.gamma.grad, dbeta=bn.beta.grad)
        finally:
            L.FUSE_BN_SUMS[0] = True
            L.BN_SUMS.clear()
    assert any("bn_bwd_reduce" in n for n in names[False])
    assert not any("bn_bwd_reduce" in n for n in names[True]), names[True]
    assert not L.BN_SUMS
    for key in res[True]:
        a, b = res[True][key], res[False][key]
        scale = b.abs().max().item() + 1e-30
        assert (a - b).abs().max().item() <= 2e-5 * scale, (key, (a - b).abs().max().item(), scale)
    assert torch.equal(res[True]["dW"], res[False]["dW"])


def test_batchnorm_backward_sums_are_not_used_when_the_gradient_has_a_second_contribution(dev):
    """Two convs read the same BatchNormalization output: autograd adds their data gradients, so neither conv's sums
    describe the gradient the BatchNormalization receives — it must run its own reduction (and give the right answer)."""
    from embeddingnet_amd import layers as L
    x = torch.randn((4, 10, 10, 32), device=dev)
    res = {}
    for fuse in (False, True):
        L.FUSE_BN_SUMS[0] = fuse
        try:
            gen = torch.Generator().manual_seed(5)
            bn = L.BatchNormalization(32, relu=True).to(dev).train()
            c1 = L.Conv2D(32, 64, 1, use_bias=False, gen=gen).to(dev)
            c2 = L.Conv2D(32, 64, 1, use_bias=False, gen=gen).to(dev)
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            a = bn(xt)
            out = c1(a) + 0.5 * c2(a)
            out.backward(torch.sin(out.detach()))
            torch.cuda.synchronize()
            names = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            assert any("bn_bwd_reduce" in n for n in names)
            res[fuse] = (xt.grad.clone(), bn.gamma.grad.clone(), bn.beta.grad.clone())
        finally:
            L.FUSE_BN_SUMS[0] = True
            L.BN_SUMS.clear()
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ hardest mining by anchor
@pytest.mark.parametrize("p,k,e,margin", [(512, 4, 64, 0.5), (300, 6, 32, 0.3), (700, 2, 16, 0.5), (130, 8, 32, 0.5), (4096, 4, 32, 0.5)])
def test_hardest_mining_one_wave_per_anchor_matches_the_rule(dev, p, k, e, margin):
    """`hardest` at N >= 1024 runs one wave per ANCHOR row (mine_hardest_anchor_kernel): every pair (i < j) of the class gets the
    first arg-max of (D[i,j] - D[i,neg]) + margin over the out-of-class columns, kept iff > 0 — the reference's rule
    (datagenerators.py:188-190, :235) in its float32 arithmetic, checked here against that rule written with torch on the same
    matrix (duplicated rows force ties; a tight cluster gives inactive pairs), and the compaction keeps the pair order."""
    from embeddingnet_amd import ops
    n = p * k
    g = torch.Generator(device=dev).manual_seed(p + k)
    cent = torch.rand((p, e), device=dev, generator=g)
    x = (cent.repeat_interleave(k, 0) + 0.35 * torch.rand((n, e), device=dev, generator=g)).abs()
    x[5 * k + 1] = x[9 * k]                      # exact duplicates across classes: ties in the arg-max
    x[7 * k] = x[11 * k + 1]
    x[: 2 * k] = x[0] + 1e-4 * torch.rand((2 * k, e), device=dev, generator=g)   # two classes on top of each other
    x[:, -1] = 0.0                                 # class 3 alone on the last axis: sqrt(2) from everybody, its pairs stay inactive
    x[3 * k: 4 * k] = 1e-3 * torch.rand((k, e), device=dev, generator=g)
    x[3 * k: 4 * k, -1] = 1.0
    x = x / x.norm(dim=1, keepdim=True)
    d = ops.pairwise_distances(x)
    trip, count, sel = ops.mine_triplets(d, p, k, margin, "hardest")
    torch.cuda.synchronize()
    # the rule, pair by pair, in torch (float32, two roundings)
    cls = torch.arange(n, device=dev) // k
    want = torch.full((p * k * (k - 1) // 2,), -1, dtype=torch.int32, device=dev)
    ppc = k * (k - 1) // 2
    pair_of = [(i, j) for i in range(k) for j in range(i + 1, k)]
    for q, (ii, jj) in enumerate(pair_of):
        rows = torch.arange(p, device=dev) * k + ii
        dap = d[rows, rows - ii + jj]
        loss = (dap[:, None] - d[rows]) + torch.tensor(margin, device=dev)       # [p, n]
        loss = torch.where(cls[None, :] == cls[rows][:, None], torch.full_like(loss, -float("inf")), loss)
        best, idx = loss.max(dim=1)
        # first maximal index (torch.max may return any tied index: take the smallest column holding the maximum)
        first = torch.where(loss == best[:, None], torch.arange(n, device=dev)[None, :], torch.full_like(idx, n)[:, None]).min(dim=1).values
        want[torch.arange(p, device=dev) * ppc + q] = torch.where(best > 0, first, torch.full_like(first, -1)).int()
    assert torch.equal(sel, want)
    live = want >= 0
    assert int(count.item()) == int(live.sum().item()) and 0 < int(count.item()) < want.numel()
    t = trip[: int(count.item())]
    pairs = torch.nonzero(live).flatten()
    cc, qq = pairs // ppc, pairs % ppc
    ii = torch.tensor([a for a, _ in pair_of], device=dev)[qq]; jj = torch.tensor([b for _, b in pair_of], device=dev)[qq]
    assert torch.equal(t[:, 0].long(), cc * k + ii) and torch.equal(t[:, 1].long(), cc * k + jj)
    assert torch.equal(t[:, 2], want[live])


@pytest.mark.parametrize("mode", ["semihard", "random_hard", "hardest"])
@pytest.mark.parametrize("p,k,e", [(512, 4, 64), (300, 6, 32), (700, 2, 16), (130, 8, 32), (2048, 4, 32)])
def test_mining_by_anchor_equals_mining_by_pair(dev, p, k, e, mode):
    """From N = 1024 on the mining kernels run one wave per anchor row (the row is read once — twice for the random rules — for
    all of the anchor's pairs) instead of one wave per pair: selections, candidate masks and the compacted triplet list are
    bit-identical (same predicates, same column order, same counter RNG per pair)."""
    import os
    from embeddingnet_amd import ops
    n = p * k
    g = torch.Generator(device=dev).manual_seed(3 * p + k)
    cent = torch.rand((p, e), device=dev, generator=g)
    x = (cent.repeat_interleave(k, 0) + 0.3 * torch.rand((n, e), device=dev, generator=g)).abs()
    x[5 * k + 1] = x[9 * k]
    x[:, -1] = 0.0
    x[3 * k: 4 * k] = 1e-3 * torch.rand((k, e), device=dev, generator=g)
    x[3 * k: 4 * k, -1] = 1.0
    x = x / x.norm(dim=1, keepdim=True)
    d = ops.pairwise_distances(x)
    out = {}
    for by_anchor in ("0", "1"):
        os.environ["EMBNET_MINE_BY_ANCHOR"] = by_anchor
        try:
            _lib.trace_reset(); _lib.trace_enable(True)
            trip, count, sel, mask = ops.mine_triplets(d, p, k, 0.5, mode, seed=1234, with_candidates=True)
            torch.cuda.synchronize()
            names = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
        finally:
            os.environ.pop("EMBNET_MINE_BY_ANCHOR", None)
        assert any("anchor" in nm for nm in names) == (by_anchor == "1"), names
        out[by_anchor] = (trip.clone(), count.clone(), sel.clone(), mask.clone())
    cnt = int(out["0"][1].item())
    assert 0 < cnt < out["0"][2].numel()
    assert torch.equal(out["0"][1], out["1"][1]) and torch.equal(out["0"][2], out["1"][2])
    assert torch.equal(out["0"][3], out["1"][3])
    assert torch.equal(out["0"][0][:cnt], out["1"][0][:cnt])


# ------------------------------------------------------------------------------------------------ pooled gradient in BN backward
@pytest.mark.parametrize("shape,act", [((6, 14, 14, 96), "swish"), ((3, 7, 9, 240), "swish"), ((5, 28, 28, 32), "relu"),
                                       ((256, 14, 14, 480), "swish")])
def test_pooled_gradient_is_added_inside_the_batchnorm_backward(dev, shape, act):
    """BatchNormalization(emit_gap=True) (the squeeze-and-excite input of an MBConv block, reference backbones.py:84-98): the
    gradient of the pooled branch is added to dy inside the two BatchNorm-backward passes (embnet_bn_bwd_gap) instead of by
    embnet_gap_bwd's pass — same summed gradient (embnet_gap_bwd's two roundings), dx / dgamma / dbeta equal to the last bits
    (the compiler contracts the apply arithmetic differently in the two kernels), no gap_bwd kernel launched."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_GAP_BN[0] = fuse
        try:
            c = shape[-1]
            bn = L.BatchNormalization(c, activation=act).to(dev).train()
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y, pooled = bn(xt, emit_gap=True)
            gate = torch.sigmoid(pooled * 3.0)                                   # a stand-in for the squeeze-excite dense layers
            out = y * gate.reshape(shape[0], 1, 1, c)
            out.backward(torch.cos(out.detach() * 2.0))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = (xt.grad.clone(), bn.gamma.grad.clone(), bn.beta.grad.clone())
        finally:
            L.FUSE_GAP_BN[0] = True
    assert any("gap_bwd" in n for n in names[False]) and not any("gap_bwd" in n for n in names[True]), names[True]
    assert any("bn_bwd_reduce4_gap" in n for n in names[True])
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 2e-6 * (b.abs().max().item() + 1e-30)


@pytest.mark.parametrize("shape", [(6, 14, 14, 96), (3, 7, 9, 240), (128, 14, 14, 480)])
def test_gate_multiply_backward_rides_on_the_batchnorm_backward(dev, shape):
    """The squeeze-and-excite chain of an MBConv block (reference backbones.py:84-98): y, pooled = BN(x); s = gate(pooled);
    out = y * s.  With lazy_scale the scaling's backward only computes the gate's gradient (embnet_channel_scale_dgate) and
    the BatchNorm backward applies dy * s itself (embnet_bn_bwd_gap's gate): no chscale_bwd / gap_bwd kernel, the same
    gradients to the last bits."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    n, c = shape[0], shape[-1]
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_GATE_BN[0] = fuse
        L.SE_BN_SUMS[0] = False                      # (the reduction pass stays: this test is about the gate multiply alone)
        try:
            gen = torch.Generator().manual_seed(2)
            bn = L.BatchNormalization(c, activation="swish").to(dev).train()
            se = L.Dense(c, c, gen=gen).to(dev)
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y, pooled = bn(xt, emit_gap=True, lazy_scale=True)
            s = L.sigmoid(se(pooled))
            out = L.channel_scale(y, s, lazy=True)
            out.backward(torch.cos(out.detach() * 2.0))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = (xt.grad.clone(), bn.gamma.grad.clone(), bn.beta.grad.clone(), se.kernel.grad.clone(), se.bias.grad.clone())
        finally:
            L.FUSE_GATE_BN[0] = True
            L.SE_BN_SUMS[0] = True
            L.GATE_PENDING.clear()
    assert any("chscale_bwd" in nm for nm in names[False])
    assert not any("chscale_bwd" in nm or "gap_bwd" in nm for nm in names[True]), names[True]
    assert any("chscale_dgate4" in nm for nm in names[True]) and not L.GATE_PENDING
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 2e-6 * (b.abs().max().item() + 1e-30)


def test_lazy_gate_with_a_second_consumer_fails_loudly(dev):
    """lazy_scale is a promise that the BatchNormalization output feeds the scaling only; a second consumer makes autograd
    add another gradient to the unscaled one — the BatchNormalization backward must refuse instead of returning a wrong dx."""
    from embeddingnet_amd import layers as L
    x = torch.randn((4, 6, 6, 32), device=dev, requires_grad=True)
    bn = L.BatchNormalization(32, activation="swish").to(dev).train()
    y, pooled = bn(x, emit_gap=True, lazy_scale=True)
    out = L.channel_scale(y, torch.sigmoid(pooled), lazy=True) + 0.5 * y          # the forbidden second use of y
    with pytest.raises(_lib.EmbnetError, match="another consumer"):
        out.sum().backward()
    L.GATE_PENDING.clear()


@pytest.mark.parametrize("shape,act", [((6, 14, 14, 96), "swish"), ((3, 7, 9, 240), "swish"), ((64, 28, 28, 144), "swish"),
                                       ((5, 10, 10, 32), "relu")])
def test_squeeze_excite_backward_needs_no_batchnorm_reduction_pass(dev, shape, act):
    """embnet_se_bn_sums: the pass that sums the gate's gradient also leaves, per image and channel, the four sums from which
    the BatchNormalization's dbeta / dgamma follow (the output gradient a' (dg s + dpool / hw) is linear in the two per-(n,c)
    factors) — no bn_bwd_reduce kernel runs, and every gradient equals the pass-by-pass chain within fp32 re-association."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    n, c = shape[0], shape[-1]
    res, names = {}, {}
    for fuse in (False, True):
        L.SE_BN_SUMS[0] = fuse
        try:
            gen = torch.Generator().manual_seed(2)
            bn = L.BatchNormalization(c, activation=act).to(dev).train()
            se = L.Dense(c, c, gen=gen).to(dev)
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y, pooled = bn(xt, emit_gap=True, lazy_scale=True)
            out = L.channel_scale(y, L.sigmoid(se(pooled)), lazy=True)
            out.backward(torch.cos(out.detach() * 2.0))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(dx=xt.grad.clone(), dgamma=bn.gamma.grad.clone(), dbeta=bn.beta.grad.clone(), dW=se.kernel.grad.clone(),
                             db=se.bias.grad.clone())
        finally:
            L.SE_BN_SUMS[0] = True
            L.GATE_PENDING.clear()
    assert any("bn_bwd_reduce4_gap" in nm for nm in names[False]) and any("chscale_dgate4" in nm for nm in names[False])
    assert not any("bn_bwd_reduce" in nm or "chscale" in nm.replace("chscale_fwd", "") for nm in names[True]), names[True]
    assert any("se_bn_sums4" in nm for nm in names[True])
    for key in res[True]:
        a, b = res[True][key], res[False][key]
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-30), (key, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("shape,act", [((6, 14, 14, 96), "swish"), ((3, 7, 9, 240), "swish"), ((64, 28, 28, 144), "swish"),
                                       ((5, 10, 10, 32), "relu")])
def test_se_gate_never_writes_the_activated_tensor(dev, shape, act):
    """BatchNormalization.se_gate: the pooled means from one pass over the BatchNormalization's input (no output written), the
    gated output act(BN(x)) * s from a second — no chscale / affine_act_gap-with-output / bn_bwd_reduce kernels — with the same
    output and gradients as the layer-by-layer chain (output to the last bit of the product, gradients within 2e-5)."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    n, c = shape[0], shape[-1]
    res, names = {}, {}
    for fuse in (False, True):
        L.SE_TWO_STAGE[0] = fuse
        try:
            gen = torch.Generator().manual_seed(2)
            bn = L.BatchNormalization(c, activation=act).to(dev).train()
            se = L.Dense(c, c, gen=gen).to(dev)
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            out = bn.se_gate(xt, lambda g: L.sigmoid(se(g)))
            out.backward(torch.cos(out.detach() * 2.0))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(out=out.detach().clone(), dx=xt.grad.clone(), dgamma=bn.gamma.grad.clone(), dbeta=bn.beta.grad.clone(),
                             dW=se.kernel.grad.clone(), db=se.bias.grad.clone(), mm=bn.moving_mean.clone())
        finally:
            L.SE_TWO_STAGE[0] = True
            L.GATE_PENDING.clear(); L.POOL_PENDING.clear()
    assert any("chscale_fwd" in nm for nm in names[False])
    assert not any("chscale" in nm or "bn_bwd_reduce" in nm for nm in names[True]), names[True]
    assert any("affine_act_scale4" in nm for nm in names[True]) and not L.POOL_PENDING
    for key in res[True]:
        a, b = res[True][key], res[False][key]
        tol = 2e-7 if key in ("out", "mm") else 2e-5
        assert (a - b).abs().max().item() <= tol * (b.abs().max().item() + 1e-30), (key, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("shape,rate", [((8, 14, 14, 80), 0.25), ((5, 7, 9, 192), 0.1), ((6, 28, 28, 40), 0.0), ((64, 14, 14, 112), 0.3)])
def test_batchnorm_dropconnect_add_in_one_pass(dev, shape, rate):
    """The tail of an MBConv block with an identity shortcut (reference backbones.py:84-98): project BatchNormalization ->
    DropConnect (whole samples) -> Add(block input).  BatchNormalization.drop_add does the three in one forward pass
    (bit-identical output: the same mask, the same roundings) and its backward applies the drop factor inside the BatchNorm
    backward — no sample_dropout / add / bn_bwd_reduce4-without-gate kernels; gradients to the last bits; the mask changes per step."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    skip0 = torch.randn(shape, device=dev)
    c = shape[-1]
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_DROP_ADD[0] = fuse
        try:
            bn = L.BatchNormalization(c).to(dev).train()
            drop = L.DropConnect(rate, seed=9).train() if rate > 0 else None
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            outs = []
            _lib.trace_reset(); _lib.trace_enable(True)
            for step in range(2):
                xt = x.clone().requires_grad_(True); sk = skip0.clone().requires_grad_(True)
                bn.gamma.grad = bn.beta.grad = None
                out = bn.drop_add(xt, sk, drop)
                out.backward(torch.cos(out.detach() * 2.0 + step))
                outs.append((out.detach().clone(), xt.grad.clone(), sk.grad.clone(), bn.gamma.grad.clone(), bn.beta.grad.clone()))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = outs
        finally:
            L.FUSE_DROP_ADD[0] = True
    assert not any("sample_dropout" in nm or nm.endswith("add_kernel") for nm in names[True]), names[True]
    assert any("affine_drop_add4" in nm for nm in names[True])
    for sa, sb in zip(res[True], res[False]):
        assert torch.equal(sa[0], sb[0]) and torch.equal(sa[2], sb[2])           # output and the skip's gradient: bit-identical
        for a, b in zip(sa[1:], sb[1:]):
            assert (a - b).abs().max().item() <= 2e-6 * (b.abs().max().item() + 1e-30)
    if rate > 0:
        dropped = [(o[0] == skip0).flatten(1).all(dim=1) for o in res[True]]     # a dropped sample's output is the skip itself
        assert int(dropped[0].sum()) + int(dropped[1].sum()) > 0 or shape[0] < 6


@pytest.mark.parametrize("shape,k,stride", [((6, 14, 14, 96), 3, 1), ((3, 15, 17, 240), 5, 2), ((5, 28, 28, 32), 5, 1),
                                            ((4, 7, 7, 1152), 3, 1), ((32, 56, 56, 144), 3, 2)])
def test_depthwise_forward_emits_the_batchnorm_statistics(dev, shape, k, stride):
    """DepthwiseConv2D(emit_stats=True) -> BatchNormalization (an MBConv block's dwconv -> bn, reference backbones.py:84-98): the
    depthwise kernel writes the per-channel sums of its output as the BatchNormalization's statistics partials (also for
    C / 4 > 256, where a workgroup covers part of the channels), no bn_stats kernel runs, and the layer's output, moving
    statistics and every gradient equal the separate-pass chain within fp32 summation order."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    c = shape[-1]
    res, names = {}, {}
    for fuse in (False, True):
        L.DW_EMIT_STATS[0] = fuse
        try:
            gen = torch.Generator().manual_seed(4)
            dw = L.DepthwiseConv2D(c, k, strides=stride, gen=gen).to(dev)
            bn = L.BatchNormalization(c, activation="swish").to(dev).train()
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y = bn(dw(xt, emit_stats=True))
            y.backward(torch.cos(y.detach() * 2.0))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(y=y.detach().clone(), dx=xt.grad.clone(), dw=dw.depthwise_kernel.grad.clone(), dgamma=bn.gamma.grad.clone(),
                             dbeta=bn.beta.grad.clone(), mm=bn.moving_mean.clone(), mv=bn.moving_variance.clone())
        finally:
            L.DW_EMIT_STATS[0] = True
    assert any("bn_stats" in nm for nm in names[False]) and not any("bn_stats" in nm for nm in names[True]), names[True]
    for key in res[True]:
        a, b = res[True][key], res[False][key]
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-30), (key, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("shape,k,stride", [((6, 14, 14, 96), 3, 1), ((3, 15, 17, 240), 5, 1), ((4, 7, 7, 1152), 3, 1),
                                            ((32, 28, 28, 144), 5, 1), ((6, 14, 14, 96), 3, 2), ((3, 15, 17, 240), 5, 2),
                                            ((2, 9, 9, 1152), 5, 2), ((16, 56, 56, 144), 5, 2), ((5, 12, 13, 32), 3, 2)])
def test_depthwise_data_gradient_emits_the_batchnorm_backward_sums(dev, shape, k, stride):
    """BatchNormalization(swish) -> DepthwiseConv2D (stride 1 or 2) (an MBConv block's expand_bn -> dwconv, reference backbones.py:84-98):
    the depthwise data gradient emits the BatchNorm-backward sums (embnet_dwconv2d_dgrad_bnsums_f32), the BatchNormalization
    backward starts at its finalize kernel — no bn_bwd_reduce kernel — and everything equals the separate-pass chain within fp32
    summation order (the depthwise gradients themselves bit for bit)."""
    from embeddingnet_amd import layers as L
    x = torch.randn(shape, device=dev)
    c = shape[-1]
    res, names = {}, {}
    for fuse in (False, True):
        L.DW_BN_SUMS[0] = fuse
        try:
            gen = torch.Generator().manual_seed(4)
            bn = L.BatchNormalization(c, activation="swish").to(dev).train()
            dw = L.DepthwiseConv2D(c, k, strides=stride, gen=gen).to(dev)
            with torch.no_grad():
                bn.gamma.copy_(torch.linspace(0.5, 1.5, c)); bn.beta.copy_(torch.linspace(-0.3, 0.3, c))
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y = dw(bn(xt))
            y.backward(torch.cos(y.detach() * 2.0))
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(dx=xt.grad.clone(), dw=dw.depthwise_kernel.grad.clone(), dgamma=bn.gamma.grad.clone(), dbeta=bn.beta.grad.clone())
        finally:
            L.DW_BN_SUMS[0] = True
            L.BN_SUMS.clear()
    assert any("bn_bwd_reduce" in nm for nm in names[False]) and not any("bn_bwd_reduce" in nm for nm in names[True]), names[True]
    assert torch.equal(res[True]["dw"], res[False]["dw"]) and not L.BN_SUMS
    for key in res[True]:
        a, b = res[True][key], res[False][key]
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-30), (key, (a - b).abs().max().item(), b.abs().max().item())


# ------------------------------------------------------------------------------------------------ conv -> ReLU -> MaxPool blocks
@pytest.mark.parametrize("shape,k,cout,pool", [((4, 21, 21, 16), 4, 64, (2, 2, 0)), ((3, 18, 15, 32), 3, 128, (2, 2, 0)),
                                               ((2, 17, 17, 8), 3, 32, (3, 2, 1)), ((32, 105, 105, 3), 10, 64, (2, 2, 0))])
def test_relu_backward_fused_into_maxpool_backward(dev, shape, k, cout, pool):
    """conv(+bias, ReLU) -> MaxPool2D (the 'simple' backbone's blocks, reference backbones.py:21-31; the last case is its first
    block at full size): with the ReLU's backward and the bias gradient folded into the pool's backward
    (embnet_maxpool_relu_bwd_colsum) the pooled output, dx and dW equal the unfused chain bit for bit (the masked gradient is the
    same number, element by element), the bias gradient within fp32 summation order — and no relu_bwd_colsum / maxpool_bwd4
    kernel is launched."""
    from embeddingnet_amd import layers as L
    x = torch.rand(shape, device=dev) - 0.3
    res, names = {}, {}
    for fuse in (False, True):
        L.FUSE_RELU_POOL[0] = fuse
        try:
            gen = torch.Generator().manual_seed(5)
            conv = L.Conv2D(shape[-1], cout, k, activation="relu", gen=gen).to(dev)
            with torch.no_grad():
                conv.bias.copy_(torch.linspace(-0.2, 0.2, cout))
            mp = L.MaxPool2D(pool[0], pool[1], zero_pad=pool[2])
            xt = x.clone().requires_grad_(True)
            _lib.trace_reset(); _lib.trace_enable(True)
            y = mp(conv(xt))
            y.backward(torch.sin(y.detach() * 3) + 0.1)
            torch.cuda.synchronize()
            names[fuse] = [r[0] for r in _lib.trace_records()]
            _lib.trace_enable(False)
            res[fuse] = dict(y=y.detach(), dx=xt.grad, dW=conv.kernel.grad, db=conv.bias.grad)
        finally:
            L.FUSE_RELU_POOL[0] = True
            L.RELU_DONE.clear()
    assert any("relu_bwd_colsum" in n for n in names[False]) and any("maxpool_bwd4" in n for n in names[False])
    assert not any("relu_bwd_colsum_kernel" in n or "maxpool_bwd4" in n for n in names[True]), names[True]
    assert any("maxpool_relu_bwd_colsum4" in n for n in names[True])
    for key in ("y", "dx", "dW"):
        assert torch.equal(res[True][key], res[False][key]), key
    a, b = res[True]["db"], res[False]["db"]
    assert (a - b).abs().max().item() <= 2e-6 * (b.abs().max().item() + 1e-30)
    assert b.abs().max().item() > 0


def test_fused_maxpool_backward_is_not_used_when_the_conv_output_has_a_second_consumer(dev):
    """The conv's output read by the pool AND by another layer: autograd hands the conv the sum of both gradients (another
    tensor), the hand-over entry is not found and the conv masks and sums itself — gradients equal the unfused run."""
    from embeddingnet_amd import layers as L
    x = torch.rand((2, 12, 12, 8), device=dev) - 0.3
    res = {}
    for fuse in (False, True):
        L.FUSE_RELU_POOL[0] = fuse
        try:
            conv = L.Conv2D(8, 16, 3, activation="relu", gen=torch.Generator().manual_seed(2)).to(dev)
            with torch.no_grad():
                conv.bias.fill_(0.05)
            xt = x.clone().requires_grad_(True)
            a = conv(xt)
            y = L.MaxPool2D()(a)
            (y.sum() * 2 + (a * a).sum()).backward()
            res[fuse] = (xt.grad, conv.kernel.grad, conv.bias.grad)
        finally:
            L.FUSE_RELU_POOL[0] = True
            L.RELU_DONE.clear()
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
