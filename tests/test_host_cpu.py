"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol the
header declares, the reference-shaped config/optimizer/model surface behaves, the product
path refuses to run without a GPU, and the data-parallel reducer works over gloo (world 2)."""
import ctypes
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from embeddingnet_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, missing
    bound = _lib.lib()                      # sets argtypes/restype from the header, checks the ABI version
    assert bound.embnet_abi_version() == 22
    assert bound.embnet_mine_max_triplets(32, 4) == 192
    assert bound.embnet_pairwise_workspace_bytes(128, 256) == 512          # row norms only: short reduction, no K split
    # the reference's default encodings_len = 4096 at a 128-row batch: 4 tiles x 128 K tiles -> K split, partial Gram slabs
    assert bound.embnet_pairwise_workspace_bytes(128, 4096) == 512 + 16 * 128 * 128 * 4
    # N = 1 024: 256 tiles of 64x64 occupy every CU -> unsplit at E <= 1 024; a long reduction gets a second workgroup per CU
    assert bound.embnet_pairwise_workspace_bytes(1024, 512) == 4096
    assert bound.embnet_pairwise_workspace_bytes(1024, 4096) == 4096 + 2 * 1024 * 1024 * 4
    assert bound.embnet_pairwise_workspace_bytes(4096, 4096) == 4 * 4096     # 1 024 tiles of 128x128: unsplit
    # simple2's Flatten -> Dense(512) at batch 32: 8 output tiles, 400 K tiles -> split; ResNet heads: not
    assert bound.embnet_dense_fwd_workspace_bytes(32, 12800, 512) == 58 * 32 * 512 * 4
    assert bound.embnet_dense_fwd_workspace_bytes(128, 512, 128) == 0
    assert bound.embnet_sumsq_chunk_elems() == 4096 and bound.embnet_optimizer_chunk_elems() == 4096


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any launch, so it is testable on the CPU box."""
    from embeddingnet_amd import _lib
    lib = _lib.lib()
    rc = lib.embnet_pairwise_dist_f32(None, 4, 4, None, 0, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.embnet_last_error()
    rc = lib.embnet_mine_triplets(1, 1, 4, 0.5, 7, 0, 1, 1, 1, None, None)
    assert rc == -1 and b"k_classes>=2" in lib.embnet_last_error()
    rc = lib.embnet_conv2d_fwd_f32(16, 16, None, 8, 1, 8, 8, 3, 3, 3, 4, 1, 0, 0, 9, 9, 0, None, None, None, 0, None, None, 0, None)
    assert rc == -1 and b"16-byte aligned" in lib.embnet_last_error()
    rc = lib.embnet_conv2d_fwd_f32(16, 16, None, 32, 1, 8, 8, 3, 3, 3, 4, 1, 0, 0, 9, 9, 0, None, None, None, 0, None, None, 0, None)
    assert rc == -1 and b"reaches outside" in lib.embnet_last_error()


def test_product_path_has_no_cpu_fallback():
    from embeddingnet_amd import _lib, ops
    from embeddingnet_amd.losses_and_accuracies import triplet_loss
    with pytest.raises(_lib.EmbnetError):
        triplet_loss(0.5)(None, torch.zeros(4, 12))
    with pytest.raises(_lib.EmbnetError):
        ops.pairwise_distances(torch.zeros(4, 8))
    src = open(os.path.join(ROOT, "embeddingnet_amd", "ops.py")).read() + \
        open(os.path.join(ROOT, "embeddingnet_amd", "layers.py")).read()
    assert "oracle" not in src, "product code must never import the oracle"


def test_parse_params_keeps_reference_schema(tmp_path):
    from embeddingnet_amd.utils import OptimizerSpec, parse_params
    cfg = tmp_path / "cfg.yml"
    cfg.write_text(textwrap.dedent("""
        MODEL:
          input_shape : [64, 64, 3]
          encodings_len: 256
          mode : 'triplet'
          distance_type : 'l1'
          backbone_name : 'simple2'
          backbone_weights : null
          freeze_backbone : False
          embeddings_normalization: True
        DATALOADER:
          dataset_path : '/tmp/none'
          validate : True
          val_ratio : 0.2
        GENERATOR:
          negatives_selection_mode : 'semihard'
          k_classes: 8
          k_samples: 4
          margin: 0.5
          batch_size : 8
          n_batches : 10
          augmentations : 'none'
        TRAIN:
          optimizer : 'radam'
          learning_rate : 0.0001
          decay_factor : 0.99
          step_size : 1
          n_epochs : 2
          plot_history : False
        ENCODINGS:
          save_encodings : True
        GENERAL:
          project_name : 'p'
          work_dir : 'work_dirs/'
    """))
    p = parse_params(str(cfg))
    assert set(p) == {"dataloader", "generator", "model", "train", "general", "encodings"}
    assert p["generator"]["input_shape"] == [64, 64, 3] and p["generator"]["augmentations"] is None
    spec = p["train"]["optimizer"]
    assert isinstance(spec, OptimizerSpec)
    w = torch.nn.Parameter(torch.zeros(3))
    from embeddingnet_amd import _lib
    from embeddingnet_amd.optimizers import KerasOptimizer
    opt = spec.build([w])
    assert isinstance(opt, KerasOptimizer) and isinstance(opt, torch.optim.Optimizer)
    assert (opt.rule, opt.eps, opt.defaults["lr"], opt.b1, opt.b2) == ("radam", 1e-7, 1e-4, 0.9, 0.999)
    assert OptimizerSpec("adam", 1e-3).build([w]).rule == "adam"
    assert OptimizerSpec("rms_prop", 1e-3).build([w]).rule == "rms_prop"
    assert OptimizerSpec("whatever", 1e-3).build([w]).rule == "sgd"         # reference utils.py:151-152
    w.grad = torch.ones(3)
    with pytest.raises(_lib.EmbnetError):                                   # the update is a HIP launch: no CPU path
        opt.step()
    # RAdam's host-side scalars: un-rectified for the first steps (sma_t < 5), rectified after
    assert opt._coefficients(1e-4, 1)[0] == 4 and opt._coefficients(1e-4, 6)[0] == 3


def test_oracle_optimizer_rules_against_torch_optim():
    """oracle/optimizers.py (parity unpinned: TF 2.2 / keras_radam rules restated) cross-checked against torch.optim on
    CPU where the two rules coincide: SGD and RMSprop exactly; Adam / RAdam up to where epsilon enters (Keras adds eps
    to sqrt(v) before the bias correction, torch after), so they agree to ~eps/|g| on O(1) gradients."""
    from oracle import optimizers as OO
    rs = np.random.RandomState(0)
    w0 = [rs.randn(5, 3), rs.randn(7)]
    grads = [[rs.randn(5, 3), rs.randn(7)] for _ in range(12)]
    for name, topt, tol in (("sgd", lambda p: torch.optim.SGD(p, lr=1e-2), 1e-13),
                            ("rms_prop", lambda p: torch.optim.RMSprop(p, lr=1e-2, alpha=0.9, eps=1e-7), 1e-13),
                            ("adam", lambda p: torch.optim.Adam(p, lr=1e-2, eps=1e-7), 2e-6),
                            ("radam", lambda p: torch.optim.RAdam(p, lr=1e-2, eps=1e-7), 2e-6)):
        wn = [a.copy() for a in w0]
        wt = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in w0]
        oo, to = OO.get_optimizer(name, 1e-2), topt(wt)
        for g in grads:
            oo.step(wn, g)
            for t, gg in zip(wt, g):
                t.grad = torch.tensor(gg)
            to.step()
        for a, t in zip(wn, wt):
            np.testing.assert_allclose(a, t.detach().numpy(), rtol=tol, atol=tol)
    # known answer, Adam step 1: m = .1 g, v = .001 g^2, lr_t = lr sqrt(.001)/.1 -> w - lr * g/(|g| + eps/sqrt(.001))... Keras form:
    w, g = [np.array([1.0])], [np.array([0.5])]
    OO.Adam(0.1).step(w, g)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w[0], 1.0 - lr_t * 0.05 / (np.sqrt(0.001 * 0.25) + 1e-7), rtol=1e-14)
    # RAdam: steps 1..5 are plain bias-corrected momentum (sma_t < 5), the 6th is rectified
    w, r = [np.array([0.0])], OO.RAdam(0.1)
    r.step(w, [np.array([2.0])])
    np.testing.assert_allclose(w[0], -0.1 * 2.0, rtol=1e-14)                # m^ = g at t = 1


def test_model_surface_matches_reference_names():
    """Attribute / constructor surface of models.py and backbones.py (no kernels run: CPU device)."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.models import SiameseNet, TripletNet
    cpu = torch.device("cpu")
    params = {"model": dict(input_shape=[64, 64, 3], encodings_len=32, mode="triplet", distance_type="l2",
                            backbone_name="simple2", backbone_weights=None, freeze_backbone=False,
                            embeddings_normalization=True, device=cpu),
              "dataloader": {}, "generator": {}, "train": {},
              "general": {"work_dir": "work_dirs/", "project_name": "x"}}
    net = TripletNet(params, training=True)
    for attr in ("params_model", "params_dataloader", "params_generator", "params_general", "params_train",
                 "base_model", "backbone_model", "model", "classification_model", "workdir_path",
                 "encoded_training_data"):
        assert hasattr(net, attr), attr
    assert net.workdir_path == os.path.join("work_dirs/", "x")
    names = set(B.keras_weights(net.base_model))
    assert {"conv1/kernel", "conv1/bias", "bn7/gamma", "bn7/moving_variance", "dense1/kernel", "dense2/bias"} <= names
    assert net.base_model.net.backbone.conv3.geometry(60, 60) == (2, 1, 1, 30, 30)     # 'same', stride 2
    assert net.base_model.net.backbone.conv3.geometry(59, 59) == (2, 2, 2, 30, 30)
    assert SiameseNet(params, training=True).model is not None
    assert TripletNet(params, training=False).model is None
    # ResNet18 parameter count (11.17 M conv + BN + 0.099 M head = 11.28 M)
    base, backbone = B.get_backbone((224, 224, 3), encodings_len=256, backbone_name="resnet18", backbone_weights=None,
                                    device=cpu)
    n = sum(p.numel() for p in base.parameters())
    assert abs(n - 11.28e6) < 0.03e6, n
    with pytest.raises(KeyError):
        B.get_backbone((64, 64, 3), backbone_name="vgg99", backbone_weights=None, device=cpu)


def test_oracle_step_runs_and_learns_on_cpu():
    """The timed CPU baseline (oracle/step.py) is a real training step: loss goes down on a fixed batch."""
    from oracle.step import ReferenceStep
    rs = np.random.RandomState(0)
    p, k = 3, 3
    cls = rs.rand(p, 64, 64, 3)
    x = np.clip(np.repeat(cls, k, axis=0) + 0.2 * rs.randn(p * k, 64, 64, 3), 0, 1).astype(np.float32)
    ref = ReferenceStep("simple2", (64, 64, 3), 32, p, k, 0.5, "hardest", lr=1e-3, optimizer="adam")
    losses = [ref.step(x)[0] for _ in range(6)]
    assert losses[-1] < losses[0]


def test_shard_classes():
    from embeddingnet_amd.parallel import shard_classes
    assert [shard_classes(256, 8, r) for r in (0, 3, 7)] == [(0, 32), (96, 32), (224, 32)]
    with pytest.raises(ValueError):
        shard_classes(30, 8, 0)
    with pytest.raises(ValueError):
        shard_classes(8, 8, 0)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from embeddingnet_amd.parallel import (GradReducer, all_reduce_mean, average_buffers, broadcast_model, init_distributed,
                                       shard_classes)
rank, world, _ = init_distributed("gloo")
assert shard_classes(8, world, rank) == (rank * 4, 4)           # whole classes per rank, as tools/train.py shards them
torch.manual_seed(100 + rank)                                   # ranks start DIFFERENT ...
class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.bn, self.b = torch.nn.Linear(20, 64), torch.nn.BatchNorm1d(64), torch.nn.Linear(64, 8)
        self.unused = torch.nn.Linear(5, 5)                     # trainable but not on the step's graph (no hook fires)
        self.frozen = torch.nn.Linear(64, 64)
        for p in self.frozen.parameters():
            p.requires_grad_(False)                             # a frozen backbone layer: not in the reducer at all
    def forward(self, x):
        return self.b(torch.relu(self.bn(self.a(x)) + self.frozen(torch.zeros(1, 64))))
model = Net()
broadcast_model(model)                                          # ... and are made identical: parameters AND buffers
state = torch.cat([t.detach().reshape(-1).float() for t in list(model.parameters()) + list(model.buffers())])
others = [torch.zeros_like(state) for _ in range(world)]
dist.all_gather(others, state)
assert all(torch.equal(o, others[0]) for o in others), "broadcast_model left the ranks different"
params = [p for p in model.parameters() if p.requires_grad]
red = GradReducer(params, bucket_bytes=2048)                    # several buckets
assert len(red.buckets) > 1
first_layout = list(red.order)
g = torch.Generator().manual_seed(100 + rank)                   # different local batch per rank
x = torch.randn(16, 20, generator=g)
for step in range(3):
    red.zero()                                                  # re-arms the bucket counters
    model(x).pow(2).mean().backward()
    red.finish()
    if step == 0:                                               # layout now follows the first backward's hook order:
        assert red.order[0] is model.b.bias or red.order[0] is model.b.weight, "last layer's gradients come first"
        assert red.order[-1] is model.unused.bias or red.order[-1] is model.unused.weight, "never-fired parameters last"
def packed():                                                   # the slots without their 16-byte alignment padding
    return torch.cat([red.flat[o:o + n] for o, n in (red._slot[p] for p in red.order)])
for p in params:                                                # every slot on a 16-byte boundary, padding stays zero
    assert red._slot[p][0] % 4 == 0 and p.grad.data_ptr() % 16 == red.flat.data_ptr() % 16
assert red.flat.numel() > sum(p.numel() for p in params), "this net has odd-sized parameters: the buffer must be padded"
assert float(red.flat.abs().sum()) == float(packed().abs().sum())
flat = packed()
# reference: mean over ranks of the local gradients, computed without the reducer
ref = Net()
ref.load_state_dict(model.state_dict())
ref(x).pow(2).mean().backward()
want = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1)
                  for q in [dict(ref.named_parameters())[n] for n in
                            [next(n for n, p in model.named_parameters() if p is o) for o in red.order]]])
dist.all_reduce(want); want /= world
assert torch.allclose(flat, want, rtol=1e-5, atol=1e-7), (flat - want).abs().max()
assert float(model.unused.weight.grad.abs().max()) == 0.0       # unused parameter: zero gradient, bucket still reduced
for p in params:
    off, n = red._slot[p]
    assert p.grad.data_ptr() == red.flat.data_ptr() + 4 * off   # grads are views of the flat buffer
others = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(others, flat)
assert all(torch.equal(o, flat) for o in others)               # every rank holds the same averaged gradient
# a caller that dropped the views (zero_grad(set_to_none=True)): the hooks copy the gradients back into the buffer
for p in params:
    p.grad = None
red.zero()
model(x).pow(2).mean().backward()
red.finish()
assert torch.allclose(packed(), flat, rtol=1e-5, atol=1e-7)
# a second backward before finish(): every parameter still counts once (no negative bucket counters, no early re-launch)
red.zero()
model(x).pow(2).mean().backward()
model(x).pow(2).mean().backward()
assert all(l >= 0 for l in red._left), red._left
red.finish()
# gradient accumulation proper: hold the collectives while the backwards add up, then exchange everything once
red.zero(); red.hold(True)
model(x).pow(2).mean().backward()
model(x).pow(2).mean().backward()
red.hold(False); red.reduce_all()
assert torch.allclose(packed(), 2 * flat, rtol=1e-5, atol=1e-7)
# hold(): a captured step's backward counts but launches nothing; reduce_all() then exchanges every bucket
red.zero(); red.hold(True)
model(x).pow(2).mean().backward()
red.finish()
assert not red._works
red.hold(False); red.reduce_all()
assert torch.allclose(packed(), flat, rtol=1e-5, atol=1e-7)
# direct(): kernels write the flat-buffer views themselves and notify the reducer (layers.GRAD_SINKS); emulated here by
# filling the views by hand for the last layer and letting autograd deliver the rest
from embeddingnet_amd import layers as L
red.direct(True)
assert set(L.GRAD_SINKS) == {{p.data_ptr() for p in params}}
local = Net(); local.load_state_dict(model.state_dict())
local(x).pow(2).mean().backward()
red.zero()
for p in model.b.parameters():
    p.requires_grad_(False)                                     # autograd no longer produces these two gradients
model(x).pow(2).mean().backward()
for p, q in zip(model.b.parameters(), local.b.parameters()):
    view, notify = L.GRAD_SINKS[p.data_ptr()]
    view.copy_(q.grad); notify(); notify()                      # (a repeated notify is ignored)
    p.requires_grad_(True)
red.finish()
assert torch.allclose(packed(), flat, rtol=1e-5, atol=1e-7)
red.close()
assert not L.GRAD_SINKS and all(p.grad is None for p in params)
model(x).pow(2).mean().backward()                               # hooks are gone: plain autograd gradients again
assert model.a.weight.grad is not None and model.a.weight.grad.data_ptr() != red.flat.data_ptr()
# scalar all-reduce used for the logged loss / plateau decisions, and the BatchNorm buffer average before a checkpoint
assert all_reduce_mean(float(rank)) == 0.5
model.bn.running_mean.fill_(float(rank))
average_buffers(model)
assert torch.all(model.bn.running_mean == 0.5)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_grad_reducer_final_bucket_is_small():
    """The gradients produced last in backward get a bucket of their own (its all-reduce is the exposed one)."""
    import torch
    from embeddingnet_amd.parallel import GradReducer
    sizes = [300_000, 200_000, 100_000, 50_000, 3_000, 2_000, 500]           # backward order after reversal below
    params = [torch.nn.Parameter(torch.zeros(n)) for n in reversed(sizes)]
    red = GradReducer(params, bucket_bytes=1 << 20, tail_bytes=32 << 10)        # 262 144 floats per bucket, 8 192-float tail
    lens = [e - s for s, e, _ in red.buckets]
    assert sum(lens) == sum(sizes) and lens[-1] == 3_000 + 2_000 + 500, lens     # the tail: the last parameters within 32 KiB
    assert lens[:-1] == [300_000, 300_000, 50_000], lens
    assert [b[2] for b in red.buckets] == [1, 2, 1, 3]
    for p in params:                                                            # views still tile the flat buffer exactly
        off, n = red._slot[p]
        assert p.grad.data_ptr() == red.flat.data_ptr() + 4 * off and n == p.numel()
    red.close()


def test_grad_reducer_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT))
    port = 29400 + os.getpid() % 500
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("ok" in o for o in outs)


def test_softmax_oracle_matches_torch_reference():
    """oracle.losses.softmax_cross_entropy (parity unpinned: TF arithmetic) vs torch's own CE on CPU."""
    import torch
    from oracle import losses as olosses
    rs = np.random.RandomState(3)
    z = rs.randn(9, 7) * 4
    t = np.eye(7)[rs.randint(0, 7, 9)]
    zt = torch.tensor(z, requires_grad=True)
    ref = torch.nn.functional.cross_entropy(zt, torch.tensor(t.argmax(1)))
    ref.backward()
    loss, acc, prob, grad = olosses.softmax_cross_entropy(z, t)
    np.testing.assert_allclose(loss, ref.item(), rtol=1e-12)
    np.testing.assert_allclose(grad, zt.grad.numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(prob.sum(1), 1.0, rtol=1e-12)
    assert acc == np.mean(z.argmax(1) == t.argmax(1))


def test_plateau_callbacks_follow_keras_semantics():
    """tools/train.py Plateau = ReduceLROnPlateau(0.1, patience 4, min_delta 1e-4) + EarlyStopping(patience 10) +
    ModelCheckpoint(save_best_only) with Keras' separate states (reference train.py:82-90)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from tools.train import Plateau
    p = Plateau()
    save, stop, lr = p.update(1.0, 0.01)
    assert save and not stop and lr == 0.01
    # improvements smaller than min_delta = 1e-4 do not reset ReduceLROnPlateau's wait, but they ARE new bests for the
    # checkpoint and for early stopping (min_delta 0)
    lrs, saves = [], []
    for i in range(4):
        save, stop, lr = p.update(1.0 - 1e-5 * (i + 1), 0.01)
        lrs.append(lr); saves.append(save)
    assert saves == [True] * 4 and not stop
    assert lrs[:3] == [0.01] * 3 and abs(lrs[3] - 0.001) < 1e-12               # 4th epoch without a >= 1e-4 improvement
    assert p.rl_wait == 0 and abs(p.scale - 0.1) < 1e-12
    # no improvement at all: early stop after 10 epochs, one more reduction every 4
    p = Plateau()
    p.update(0.5, 1.0)
    out = [p.update(0.6, 1.0) for _ in range(10)]
    assert [o[1] for o in out] == [False] * 9 + [True]
    assert [abs(o[2] - 0.1) < 1e-12 for o in out] == [i in (3, 7) for i in range(10)]
    assert not any(o[0] for o in out)
    assert abs(p.scale - 0.01) < 1e-12


def test_lr_schedule_overrides_plateau_as_in_the_reference(monkeypatch):
    """Reference train.py:80-83: the one-argument LearningRateScheduler sets lr = lr0 * decay^floor(epoch/step) at every
    epoch begin, discarding what ReduceLROnPlateau set at the previous epoch end.  Reproduced by default;
    TRAIN.plateau_persistent multiplies the schedule by the accumulated plateau factor instead."""
    from tools.train import Plateau
    lr0, decay, step = 0.1, 0.5, 2

    def run(persistent):
        pl, used = Plateau(persistent=persistent), []
        for epoch in range(7):
            lr = lr0 * decay ** (epoch // step) * (pl.scale if persistent else 1.0)
            used.append(lr)
            pl.update(1.0, lr)                                   # a flat monitor: a reduction after epochs 4 (0-based)
        return used
    assert run(False) == [lr0 * decay ** (e // step) for e in range(7)]
    got = run(True)
    assert got[:5] == [lr0 * decay ** (e // step) for e in range(5)]
    assert abs(got[5] - lr0 * decay ** 2 * 0.1) < 1e-15 and abs(got[6] - lr0 * decay ** 3 * 0.1) < 1e-15


def test_apply_gpu_ids(monkeypatch):
    """GENERAL.gpu_ids (reference train.py:121-133): visible devices + one watched process per listed GPU (launch.spawn)."""
    import tools.train as T
    from embeddingnet_amd import launch
    calls = []
    monkeypatch.setattr(launch, "spawn", lambda n, argv, **kw: calls.append((n, argv)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.setattr(sys, "argv", ["tools/train.py", "cfg.yml", "--max_epochs", "1"])
    T.apply_gpu_ids(None)
    T.apply_gpu_ids("")
    assert not calls and "HIP_VISIBLE_DEVICES" not in os.environ
    T.apply_gpu_ids("2")                                         # one id: just the visibility mask
    assert os.environ["HIP_VISIBLE_DEVICES"] == "2" and not calls
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    with pytest.raises(SystemExit) as e:
        T.apply_gpu_ids("0, 3,5")
    assert e.value.code == 0 and os.environ["HIP_VISIBLE_DEVICES"] == "0,3,5"
    n, argv = calls[0]
    assert n == 3 and argv[0] == sys.executable and argv[1].endswith("tools/train.py") and argv[-3:] == ["cfg.yml", "--max_epochs", "1"]
    # under a launcher the world is the launcher's: nothing is re-spawned
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")
    T.apply_gpu_ids("0,1,2")
    assert not calls


_RANK_SCRIPT = """
import os, sys, time
rank = int(os.environ["RANK"])
assert os.environ["WORLD_SIZE"] == sys.argv[1] and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[2]
if mode == "ok":
    print("line from rank", rank, flush=True)
elif mode == "fail1":                       # rank 1 dies at once; the others would wait "in a collective" for a minute
    if rank == 1:
        sys.exit(7)
    time.sleep(60)
elif mode == "hang":
    time.sleep(60)
elif mode == "port":                        # the first world loses the race for its port, the second runs
    marker = sys.argv[3]
    if not os.path.exists(marker):
        if rank == 0:
            open(marker, "w").write(os.environ["MASTER_PORT"])
            sys.exit(98)
        time.sleep(60)
    print("second world on port", os.environ["MASTER_PORT"], "first was", open(marker).read(), flush=True)
"""


def test_launch_spawn_watches_every_rank(tmp_path):
    """embeddingnet_amd/launch.py: all ranks succeed -> 0 and rank 0's stdout comes through; one rank dies -> the others are
    terminated and ITS code is returned within seconds (not after the sleepers' minute); a parent-side time limit; a lost
    race for the rendezvous port (exit code 98) restarts the world on a new port."""
    import time
    from embeddingnet_amd import launch
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    run = lambda n, *a, **kw: launch.spawn(n, [sys.executable, str(script), str(n)] + list(a), log=lambda *m: None, **kw)
    out = subprocess.run([sys.executable, "-c",
                          f"import sys; sys.path.insert(0, {ROOT!r}); from embeddingnet_amd import launch; "
                          f"sys.exit(launch.spawn(3, [sys.executable, {str(script)!r}, '3', 'ok']))"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "line from rank 0"      # the other ranks' stdout is not relayed
    t0 = time.time()
    assert run(4, "fail1") == 7
    assert time.time() - t0 < 20
    t0 = time.time()
    assert run(2, "hang", timeout_s=1.5) == launch.EXIT_TIMEOUT
    assert time.time() - t0 < 20
    marker = tmp_path / "first_port"
    assert run(2, "port", str(marker)) == 0 and marker.exists()
    assert run(2, "port", str(tmp_path / "never"), retries=0) == launch.EXIT_PORT_IN_USE


def test_pin_to_gpu_numa_reads_the_topology(tmp_path, monkeypatch):
    """launch.gpu_cpu_sets / pin_to_gpu_numa on a fake sysfs tree: KFD GPU nodes in node order -> DRM render minor ->
    local_cpulist; CPU nodes skipped; a device without NUMA information (numa_node -1) and HIP_VISIBLE_DEVICES remapping."""
    from embeddingnet_amd import launch
    kfd, drm = tmp_path / "kfd", tmp_path / "drm"
    have = sorted(os.sched_getaffinity(0))
    lists = [f"{have[0]}", f"{have[-1]}", ""]
    spec = [(0, 0, None), (1, 0, None),                          # two CPU nodes
            (2, 304, (128, 0, lists[0])), (3, 304, (129, 1, lists[1])), (4, 304, (130, -1, lists[2]))]
    for node, simd, gpu in spec:
        d = kfd / str(node)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if gpu else 64}\nsimd_count {simd}\ndrm_render_minor {gpu[0] if gpu else 0}\n")
        if gpu:
            dev = drm / f"renderD{gpu[0]}" / "device"
            dev.mkdir(parents=True)
            (dev / "numa_node").write_text(f"{gpu[1]}\n")
            (dev / "local_cpulist").write_text(gpu[2] + "\n")
    sets = launch.gpu_cpu_sets(str(kfd), str(drm))
    assert sets == [{have[0]}, {have[-1]}, None]
    assert launch._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    before = os.sched_getaffinity(0)
    try:
        assert "no NUMA information" in launch.pin_to_gpu_numa(2, sets) and os.sched_getaffinity(0) == before
        msg = launch.pin_to_gpu_numa(1, sets)
        assert "pinned to 1 cores local to device 1" in msg and os.sched_getaffinity(0) == {have[-1]}
        os.sched_setaffinity(0, before)
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")          # local rank 1 is physical device 0
        launch.pin_to_gpu_numa(1, sets)
        assert os.sched_getaffinity(0) == {have[0]}
        os.sched_setaffinity(0, before)
        monkeypatch.setenv("EMBNET_PIN", "0")
        assert "EMBNET_PIN=0" in launch.pin_to_gpu_numa(0, sets) and os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)


def test_bench_rank_that_dies_in_startup_ends_the_world_at_once():
    """`python bench.py --gpus 2` where rank 1 exits during start-up (EMBNET_TEST_FAIL_RANK): rank 0 is waiting in the
    rendezvous for it; the parent must return rank 1's code within seconds, not at the distributed timeout."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(EMBNET_TEST_FAIL_RANK="1", EMBNET_DIST_BACKEND="gloo")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert time.time() - t0 < 90                                 # (the first `import torch` of a fresh container takes a while)
    assert "rank 1 exited with code 3" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_optimizer_state_roundtrip(tmp_path):
    """save_optimizer_state / load_optimizer_state: Adam moments, the step count and extras keyed by Keras weight names."""
    from embeddingnet_amd.optimizers import KerasOptimizer, load_optimizer_state, save_optimizer_state
    a, b = torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))
    named = {"dense/kernel": a, "dense/bias": b}
    opt = KerasOptimizer([a, b], "adam", 1e-3)
    for p in (a, b):
        s1, s2 = opt._slots(p)
        s1.copy_(torch.randn_like(p)); s2.copy_(torch.rand_like(p))
    opt.iterations = 17
    save_optimizer_state(str(tmp_path / "x.opt.npz"), opt, named, extra={"epoch": 3})
    a2, b2 = torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5))
    opt2 = KerasOptimizer([a2, b2], "adam", 1e-3)
    extra = load_optimizer_state(str(tmp_path / "x.opt.npz"), opt2, {"dense/kernel": a2, "dense/bias": b2})
    assert opt2.iterations == 17 and int(extra["epoch"]) == 3
    for p, q in ((a, a2), (b, b2)):
        assert torch.equal(opt.state[p]["slot1"], opt2.state[q]["slot1"]) and torch.equal(opt.state[p]["slot2"], opt2.state[q]["slot2"])
    from embeddingnet_amd import _lib
    with pytest.raises(_lib.EmbnetError):
        load_optimizer_state(str(tmp_path / "x.opt.npz"), KerasOptimizer([a2], "rms_prop", 1e-3), {"dense/kernel": a2})


def test_bench_gpus_n_spawns_its_own_ranks_and_relays_the_worst_exit_code():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts two ranks (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set) before touching any GPU and returns the worst child exit code.  Without a GPU every rank joins the
    gloo world and then stops with the product's 'needs an MI355X' message: a non-zero exit, no JSON line, both ranks heard."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.stderr.count("needs an MI355X") == 2, out.stderr[-2000:]
