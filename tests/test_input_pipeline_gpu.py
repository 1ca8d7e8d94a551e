"""GPU tests of the input pipeline (reference datagenerators.py:145-156, 202-218; train.py:172-177):
the uint8 -> float32 kernel against the reference's float32 `x / 255.`, the HBM-resident image store and the prefetcher
against the sequential sample_batch() they replace — bit for bit — and a training run of tools/train.py on a JPEG tree."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from embeddingnet_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    """6 classes x 9 JPEG / PNG files of 40x33 random pixels (so decode + resize matter)."""
    from PIL import Image
    root = tmp_path_factory.mktemp("images")
    rs = np.random.RandomState(0)
    for ci in range(6):
        os.makedirs(root / f"class{ci}")
        for i in range(9):
            arr = (rs.rand(33, 40, 3) * 255).astype(np.uint8)
            Image.fromarray(arr).save(str(root / f"class{ci}" / f"im{i}.{'jpg' if i % 3 else 'png'}"), quality=90)
    return root


def _gen(tree, shape, k_classes=4, k_samples=3):
    from embeddingnet_amd.datagenerators import ENDataLoader, TripletsDataGenerator
    dl = ENDataLoader(str(tree), validate=False)
    return TripletsDataGenerator(embedding_model=None, class_files_paths=dl.train_data, class_names=dl.class_names,
                                 input_shape=list(shape), k_classes=k_classes, k_samples=k_samples, margin=0.5,
                                 negatives_selection_mode="semihard")


@pytest.mark.parametrize("n,h,w,pad", [(5, 12, 16, None), (3, 7, 5, None), (4, 12, 16, 4), (2, 105, 105, None), (1, 1, 1, 4)])
def test_u8_to_f32_is_the_reference_division(dev, n, h, w, pad):
    """embnet_u8_to_f32: every one of the 256 byte values divided by 255 in float32, with and without a gather index, odd
    sizes (scalar path) and channel padding."""
    from embeddingnet_amd.input_pipeline import u8_to_f32
    rs = np.random.RandomState(n * 10 + h)
    src = rs.randint(0, 256, size=(n + 3, h, w, 3)).astype(np.uint8)
    flat = src.reshape(-1)
    flat[: min(256, flat.size)] = np.arange(min(256, flat.size), dtype=np.uint8)       # every byte value
    d = torch.from_numpy(src).to(dev)
    want = src.astype(np.float32) / np.float32(255.)
    out = u8_to_f32(d, None, n, pad_to=pad).cpu().numpy()
    assert out.shape == (n, h, w, pad or 3)
    assert np.array_equal(out[..., :3], want[:n]) and (pad is None or not out[..., 3:].any())
    idx = rs.permutation(n + 3)[:n].astype(np.int32)
    out = u8_to_f32(d, torch.from_numpy(idx).to(dev), n, pad_to=pad).cpu().numpy()
    assert np.array_equal(out[..., :3], want[idx])
    rc = _lib.lib().embnet_u8_to_f32(d.data_ptr(), None, 0, h * w, 3, 3, 255.0, d.data_ptr(), _lib.stream())
    assert rc != 0 and b"u8_to_f32" in _lib.lib().embnet_last_error()


@pytest.mark.parametrize("kind", ["store", "prefetch", "prefetch-threads"])
def test_feeder_equals_sample_batch(dev, tree, kind, monkeypatch):
    """The HBM-resident store and the prefetcher deliver, batch after batch, exactly the tensor the sequential
    sample_batch() -> torch.from_numpy -> .to(device) path delivered (same np.random stream, same float32 values)."""
    shape = (24, 20, 3)                                                    # a resize of the 40x33 files
    gen = _gen(tree, shape)
    np.random.seed(3)
    want = [gen.sample_batch() for _ in range(14)]
    monkeypatch.setenv("EMBNET_IMAGE_STORE", "1" if kind == "store" else "0")
    monkeypatch.setenv("EMBNET_DECODE_PROCESSES", "0" if kind == "prefetch-threads" else "1")
    kind = kind.split("-")[0]
    np.random.seed(3)
    feeder = gen.feeder(dev, depth=4, workers=3)
    assert feeder.kind.startswith(kind)
    try:
        assert kind == "store" or ("processes" in feeder.kind) == (os.environ["EMBNET_DECODE_PROCESSES"] == "1")
        for w in want[: 14 - (4 if kind == "prefetch" else 0) - 1]:
            got = feeder.next()
            assert got.dtype == torch.float32 and got.is_cuda and np.array_equal(got.cpu().numpy(), w)
    finally:
        feeder.close()


def test_train_cli_on_a_jpeg_tree(dev, tree, tmp_path):
    """tools/train.py on real files (directory loader -> feeder -> fused step): two epochs run, the log names the input
    pipeline and reports images/s, the loss is finite."""
    cfg = tmp_path / "cfg.yml"
    cfg.write_text(f"""
MODEL:
  input_shape : [32, 32, 3]
  encodings_len: 32
  mode : 'triplet'
  distance_type : 'l2'
  backbone_name : 'simple2'
  backbone_weights : null
  freeze_backbone : False
  embeddings_normalization: True
DATALOADER:
  dataset_path : '{tree}'
  validate : False
  val_ratio : 0.2
GENERATOR:
  negatives_selection_mode : 'hardest'
  k_classes: 4
  k_samples: 3
  margin: 0.5
  batch_size : 8
  n_batches : 6
  augmentations : 'none'
TRAIN:
  optimizer : 'adam'
  learning_rate : 0.001
  decay_factor : 0.5
  step_size : 1
  n_epochs : 2
  plot_history : False
ENCODINGS:
  save_encodings : False
GENERAL:
  project_name : 'jpeg_tree'
  work_dir : '{tmp_path}/work/'
""")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train.py"), str(cfg)], capture_output=True, text=True,
                         timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-2000:]
    assert "input pipeline: store" in out.stdout and "DeviceImageStore: 54 images" in out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("Epoch ")]
    assert len(lines) == 2 and all("images/s" in l and "nan" not in l for l in lines), out.stdout[-2000:]
