"""GPU: the reference-shaped generator contract and the tools/train.py counterpart end to end."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_triplets_generator_contract():
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.datagenerators import SyntheticDataLoader, TripletsDataGenerator
    from oracle import mining as omining
    from oracle import pairwise as opair
    dev = torch.device("cuda:0")
    base, _ = B.get_backbone((64, 64, 3), encodings_len=32, backbone_name="simple2", backbone_weights=None, device=dev)
    data = SyntheticDataLoader(10, 12, (64, 64, 3), validate=False)
    gen = TripletsDataGenerator(embedding_model=base, class_files_paths=data.train_data, class_names=data.class_names,
                                n_batches=7, input_shape=[64, 64, 3], k_classes=5, k_samples=3, margin=0.5,
                                negatives_selection_mode="hardest")
    assert len(gen) == 7
    np.random.seed(0)
    (a, p, n), targets = gen[0]
    t = len(targets)
    assert a.shape == p.shape == n.shape == (t, 64, 64, 3) and 1 <= t <= 15 and torch.all(targets == 1)
    # same sampling stream -> same batch; oracle mining on the model's inference embeddings gives the same triplets
    np.random.seed(0)
    batch = gen.sample_batch()
    emb = base.predict(batch)
    want = omining.mine_triplets(opair.pairwise_distances(emb), 5, 3, 0.5, "hardest")["triplets"]
    assert t == len(want)
    assert np.array_equal(a.cpu().numpy(), batch[want[:, 0]]) and np.array_equal(n.cpu().numpy(), batch[want[:, 2]])


def test_train_cli_synthetic(tmp_path):
    cfg = open(os.path.join(ROOT, "configs", "simple2_synthetic.yml")).read().replace("work_dirs/", str(tmp_path) + "/")
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(cfg)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train.py"), str(cfg_path), "--synthetic", "10",
                          "--max_epochs", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch 3/3" in out.stdout and "saving model" in out.stdout
    wdir = tmp_path / "simple2_synthetic"
    assert any(f.startswith("epoch_") for f in os.listdir(wdir / "weights"))
    hist = np.load(wdir / "plots" / "history.npz")
    assert len(hist["loss"]) == 3 and hist["loss"][-1] < hist["loss"][0]


def test_train_cli_softmax_pretraining(tmp_path):
    """SOFTMAX_PRETRAINING in the config -> backbone pre-training runs before the triplet stage (train.py:164-170)."""
    cfg = open(os.path.join(ROOT, "configs", "simple2_softmax_synthetic.yml")).read().replace("work_dirs/",
                                                                                              str(tmp_path) + "/")
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(cfg)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train.py"), str(cfg_path), "--synthetic", "10",
                          "--max_epochs", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "softmax pre-training epoch 2/2" in out.stdout and "Epoch 2/2" in out.stdout
    assert os.listdir(tmp_path / "simple2_synthetic" / "pretraining_model" / "weights")


def test_train_cli_softmax_pretraining_two_ranks(tmp_path):
    """The same config as a 2-process data-parallel run (both ranks on the box's one GPU over gloo; RCCL refuses two ranks
    per device): pre-training runs on EVERY rank under the gradient reducer — nobody waits in a collective for rank 0 —
    then the triplet stage; rank 0 writes the checkpoints, and the ranks end with identical weights."""
    cfg = open(os.path.join(ROOT, "configs", "simple2_softmax_synthetic.yml")).read().replace("work_dirs/",
                                                                                              str(tmp_path) + "/")
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(cfg)
    env = dict(os.environ, EMBNET_DIST_BACKEND="gloo", EMBNET_DUMP_FINAL_WEIGHTS=str(tmp_path / "final_rank"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "tools", "train.py"),
                          str(cfg_path), "--synthetic", "10", "--max_epochs", "2"], capture_output=True, text=True, timeout=900,
                         env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "softmax pre-training epoch 2/2" in out.stdout and "Epoch 2/2" in out.stdout
    assert os.listdir(tmp_path / "simple2_synthetic" / "pretraining_model" / "weights")
    w0, w1 = np.load(str(tmp_path / "final_rank") + "0.npz"), np.load(str(tmp_path / "final_rank") + "1.npz")
    assert set(w0.files) == set(w1.files) and len(w0.files) > 10
    for k in w0.files:
        if "moving_" in k:
            continue                                   # BatchNorm statistics are local to a rank (as the reference: no SyncBN)
        assert np.array_equal(w0[k], w1[k]), k


def test_grad_reducer_over_rccl_single_rank():
    """The DP reducer on the real backend (nccl = RCCL), world size 1: hooks fire, buckets are
    all-reduced asynchronously on RCCL's stream, finish() orders them before the optimizer, and the
    step equals the reducer-less step bit for bit."""
    script = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from embeddingnet_amd import backbones as B
from embeddingnet_amd.parallel import GradReducer, init_distributed
from embeddingnet_amd.train_step import TripletTrainer
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
x = torch.rand((16, 64, 64, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
res = []
for use in (False, True):
    base, _ = B.get_backbone((64, 64, 3), encodings_len=32, backbone_name="resnet18", backbone_weights=None, seed=3, device=dev)
    params = [p for p in base.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.1)
    red = GradReducer(params, bucket_bytes=1 << 20, always_reduce=True) if use else None
    if use:
        assert len(red.buckets) > 4
    tr = TripletTrainer(base, opt, 4, 4, margin=0.5, negatives_selection_mode="hardest", reducer=red)
    losses = [tr.step(x).item() for _ in range(3)]
    res.append((losses, torch.cat([p.detach().reshape(-1) for p in params]).clone()))
assert res[0][0] == res[1][0], (res[0][0], res[1][0])
assert torch.equal(res[0][1], res[1][1])
dist.destroy_process_group()
print("reducer ok", res[0][0])
''' % ROOT
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "reducer ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_bench_under_torchrun_single_rank():
    """bench.py through the driver's launch line (torch.distributed.run, 1 rank) prints one JSON line."""
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29544", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--backbone", "simple2",
                          "--image", "64", "--k-classes", "8"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] in ("mfma", "hbm") and d["roofline"]["achieved"] > 0


def _run_cli(cfg_path, *extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train.py"), str(cfg_path), "--synthetic", "10",
                          *extra], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def test_train_cli_resume_from(tmp_path):
    """--resume_from (reference train.py:156-157): the checkpoint a run wrote restores every weight bit for bit, and a
    resumed run starts from it (its first-epoch loss is the trained model's, not a fresh model's)."""
    from embeddingnet_amd.backbones import keras_weights
    from embeddingnet_amd.models import TripletNet
    from embeddingnet_amd.utils import parse_params
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(open(os.path.join(ROOT, "configs", "simple2_synthetic.yml")).read().replace("work_dirs/", str(tmp_path) + "/"))
    first = _run_cli(cfg_path, "--max_epochs", "3")
    wdir = tmp_path / "simple2_synthetic" / "weights"
    ckpt = str(wdir / sorted(os.listdir(wdir))[-1])
    resumed = _run_cli(cfg_path, "--max_epochs", "1", "--resume_from", ckpt)
    # the optimizer's slots and step count travel with the checkpoint (Keras' load_model restores the optimizer too)
    assert os.path.exists(str(tmp_path / "simple2_synthetic" / "optimizer" / os.path.basename(ckpt)))
    # (3 epochs x 20 batches when the best epoch was the last one; at least one epoch's worth in any case)
    it = int(resumed.split("resumed optimizer state")[1].split("iterations ")[1].split(",")[0])
    assert it >= 20 and it % 20 == 0, it
    loss_of = lambda text, epoch: float(text.split(f"Epoch {epoch}/")[1].split("loss ")[1].split()[0])
    assert loss_of(resumed, 1) < 0.8 * loss_of(first, 1), (loss_of(resumed, 1), loss_of(first, 1))
    cfg = parse_params(str(cfg_path))
    cfg["model"]["device"] = torch.device("cuda:0")
    cfg["model"]["seed"] = 99                                   # different initial weights than the checkpointed run
    net = TripletNet(cfg, training=True)
    net.load_model(ckpt)
    saved = np.load(ckpt)
    for k, v in keras_weights(net.base_model).items():
        assert np.array_equal(v.detach().cpu().numpy(), saved[k]), k


def test_train_cli_siamese_l1_checkpoints_the_whole_model(tmp_path):
    """mode 'siamese' with the 'l1' head (the one the reference's loss dict matches, train.py:118): trains, and the
    checkpoint holds the distance head 'output_siamese' and the classification head 'output_img' next to the base model."""
    text = open(os.path.join(ROOT, "configs", "simple2_synthetic.yml")).read().replace("work_dirs/", str(tmp_path) + "/")
    text = text.replace("mode : 'triplet'", "mode : 'siamese'").replace("distance_type : 'l2'", "distance_type : 'l1'")
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(text)
    out = _run_cli(cfg_path, "--max_epochs", "2")
    assert "Epoch 2/2" in out and "saving model" in out
    wdir = tmp_path / "simple2_synthetic" / "weights"
    saved = np.load(str(wdir / sorted(os.listdir(wdir))[-1]))
    assert {"output_siamese/kernel", "output_siamese/bias", "output_img/kernel", "dense2/kernel", "bn7/moving_mean"} <= set(saved.files)


def test_frozen_backbone_batchnorm_runs_in_inference_mode():
    """freeze_backbone (reference backbones.py:106-108 sets trainable=False on all but the last two layers): a frozen
    Keras BatchNormalization normalises with its moving statistics and does not update them; the last BN stays live."""
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import layers as L
    dev = torch.device("cuda:0")
    base, backbone = B.get_backbone((64, 64, 3), encodings_len=32, backbone_name="resnet18", backbone_weights=None,
                                    freeze_backbone=True, seed=1, device=dev)
    bns = [m for m in backbone.modules() if isinstance(m, L.BatchNormalization)]
    last = backbone.net.bn1
    with torch.no_grad():
        for m in bns:
            m.moving_mean.uniform_(-0.1, 0.1); m.moving_variance.uniform_(0.5, 1.5)
    before = [(m.moving_mean.clone(), m.moving_variance.clone()) for m in bns]
    base.train()
    assert all((not m.training) == (m is not last) for m in bns)
    x = torch.rand((8, 64, 64, 3), device=dev)
    emb = base(x)
    emb.sum().backward()
    for m, (mm, mv) in zip(bns, before):
        changed = not (torch.equal(m.moving_mean, mm) and torch.equal(m.moving_variance, mv))
        assert changed == (m is last), "only the un-frozen last BatchNormalization updates its moving statistics"
    trainable = {n for n, p in base.named_parameters() if p.requires_grad}
    assert all(("bn1." in n and "stage" not in n) or "head" in n for n in trainable), trainable
    assert all(p.grad is not None for n, p in base.named_parameters() if p.requires_grad)
    base.eval()
    with torch.no_grad():
        ref = base(x)
    # the frozen part computes the same thing in both modes; only the last BN differs (batch vs moving statistics)
    feats = {}
    h = backbone.net.stage4_unit2.register_forward_hook(lambda mod, i, o: feats.setdefault(len(feats), o.detach().clone()))
    base.train(); base(x); base.eval()
    with torch.no_grad():
        base(x)
    h.remove()
    assert torch.equal(feats[0], feats[1]) and ref.shape == emb.shape


def test_bench_two_ranks_over_gloo_on_one_gpu():
    """The N > 1 path of bench.py end to end on GPU tensors (rank-0 broadcast, per-rank class shards, bucketed gradient
    all-reduce from autograd hooks, max-over-ranks timing) with two ranks sharing this box's one GPU over gloo
    (EMBNET_DIST_BACKEND=gloo: RCCL refuses two ranks on one device; the driver's 8-GPU run uses nccl = RCCL)."""
    import json
    env = dict(os.environ, EMBNET_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29546", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--backbone", "resnet18",
                          "--image", "64", "--k-classes", "8"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_batch"] == 2 * 8 * 4
    assert d["config"]["parallelism"] == "dp2" and np.isfinite(d["config"]["final_loss"])


def test_train_cli_graph_mode_matches_eager(tmp_path):
    """EMBNET_GRAPH=1: tools/train.py replays the training step as a HIP graph (LR schedule applied through the
    device-resident optimizer scalars); the logged losses are those of the eager run, digit for digit."""
    hist = []
    for g in ("0", "1"):
        wd = tmp_path / f"g{g}"
        cfg = open(os.path.join(ROOT, "configs", "simple2_synthetic.yml")).read().replace("work_dirs/", str(wd) + "/")
        cfg_path = tmp_path / f"cfg{g}.yml"
        cfg_path.write_text(cfg)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train.py"), str(cfg_path), "--synthetic", "10",
                              "--max_epochs", "3"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, EMBNET_GRAPH=g))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert "graph capture failed" not in out.stderr
        h = np.load(wd / "simple2_synthetic" / "plots" / "history.npz")
        hist.append((h["loss"], h["val_loss"]))
    assert np.array_equal(hist[0][0], hist[1][0]) and np.array_equal(hist[0][1], hist[1][1])
