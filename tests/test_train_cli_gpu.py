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
