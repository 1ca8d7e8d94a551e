"""Every library launch of one traced training step, in order: name, microseconds, algorithmic work (debug / sizing aid).
  python tools/exp/step_launch_list.py resnet18 128 224 [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import _lib
from embeddingnet_amd.backbones import get_backbone
from embeddingnet_amd.train_step import TripletTrainer
from embeddingnet_amd.optimizers import KerasOptimizer

name, batch, image = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
flt = sys.argv[4] if len(sys.argv) > 4 else ""
dev = torch.device("cuda:0")
base, _ = get_backbone((image, image, 3), encodings_len=256, backbone_name=name, backbone_weights=None, seed=0, device=dev)
base.train()
tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "radam", 1e-4), k_classes=batch // 4, k_samples=4, margin=0.5,
                    negatives_selection_mode="hardest", graph=False)
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand((batch, image, image, 3), device=dev, generator=g) for _ in range(4)]
for i in range(6):
    tr.step(xs[i % 4])
torch.cuda.synchronize()
acc = {}
for rep in range(5):
    _lib.trace_reset(); _lib.trace_enable(True)
    tr.step(xs[rep % 4])
    recs = _lib.trace_records()
    _lib.trace_enable(False)
    for i, r in enumerate(recs):
        acc.setdefault(i, []).append(r)
tot = 0.0
for i in sorted(acc):
    rs = acc[i]
    ms = sorted(r[1] for r in rs)[len(rs) // 2]
    n, _, work, unit, nbytes = rs[0]
    tot += ms
    if flt in n:
        rate = (work / (ms * 1e-3) / 1e12) if unit == 0 else (work / (ms * 1e-3) / 1e9)
        print("%4d %-100s %8.1f us  %10.3f %s  %8.1f %s" % (i, n[:100], ms * 1e3, work / 1e9, "GFLOP" if unit == 0 else "GB   ", rate, "TF/s" if unit == 0 else "GB/s"))
print("total %.3f ms over %d launches" % (tot, len(acc)))
