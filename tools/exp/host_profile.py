#!/usr/bin/env python3
"""Where the host time of a training step goes (cProfile of TripletTrainer.step at 32x32 images: GPU work negligible)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from embeddingnet_amd import backbones as B
from embeddingnet_amd.optimizers import RAdam
from embeddingnet_amd.train_step import TripletTrainer
dev = torch.device("cuda:0")
SZ = int(os.environ.get("SZ", "32"))
base, _ = B.get_backbone((SZ, SZ, 3), encodings_len=256, backbone_name="resnet18", backbone_weights=None, device=dev)
opt = RAdam([p for p in base.parameters() if p.requires_grad], 1e-4)
tr = TripletTrainer(base, opt, 32, 4, margin=0.5, negatives_selection_mode="hardest")
x = torch.rand((128, SZ, SZ, 3), device=dev)
for _ in range(10): tr.step(x)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(12): tr.step(x)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
