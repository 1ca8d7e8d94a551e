"""Loss trajectories of the `simple` backbone under TripletTrainer: three products vs six terms, eager vs captured (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import layers as L
from embeddingnet_amd.backbones import get_backbone
from embeddingnet_amd.train_step import TripletTrainer
from embeddingnet_amd.optimizers import KerasOptimizer

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "simple"
shape = (105, 105, 3) if name == "simple" else (64, 64, 3)


MINING = sys.argv[2] if len(sys.argv) > 2 else "hardest"
LR = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4


def run(f16, graph, steps=40):
    L.CONV_F16[0] = f16
    torch.manual_seed(0)
    base, _ = get_backbone(shape, encodings_len=256, backbone_name=name, backbone_weights=None, seed=3, device=dev)
    base.train()
    tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "radam", LR), k_classes=8, k_samples=4, margin=0.5,
                        negatives_selection_mode=MINING, graph=graph)
    g = torch.Generator().manual_seed(9)
    out = []
    for i in range(steps):
        x = torch.rand((32,) + shape, generator=g).to(dev)
        out.append((float(tr.step(x)), int(tr.last_triplets[1][0].item())))
    return out


for f16 in (False, True):
    for graph in (False, True):
        l = run(f16, graph)
        print("f16=%d graph=%d" % (f16, graph), " ".join("%.4f/%d" % v for v in l))
