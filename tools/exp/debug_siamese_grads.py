"""Which parameter gradients of one Siamese backward differ between the plain loop and SiameseTrainer's eager step (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import layers as L
from embeddingnet_amd.losses_and_accuracies import contrastive_loss
from embeddingnet_amd.models import SiameseNet
from embeddingnet_amd.optimizers import KerasOptimizer
from embeddingnet_amd.train_step import SiameseTrainer

dev = torch.device("cuda", 0)


def net():
    torch.manual_seed(11)
    return SiameseNet({"model": dict(input_shape=[64, 64, 3], encodings_len=32, mode="siamese", distance_type="l2", backbone_name="resnet18",
                                     backbone_weights=None, freeze_backbone=False, embeddings_normalization=True, device=dev, seed=4),
                       "dataloader": {}, "generator": {}, "train": {}, "general": {"work_dir": "work_dirs/", "project_name": "t"}}, training=True)


gen = torch.Generator(device=dev).manual_seed(3)
x1, x2 = torch.rand((8, 64, 64, 3), device=dev, generator=gen), torch.rand((8, 64, 64, 3), device=dev, generator=gen)
y = (torch.arange(8, device=dev) < 4).float().reshape(-1, 1)

a = net(); a.model.train()
loss = contrastive_loss(y, a.model([x1, x2])[0]); loss.backward()
ga = {n: p.grad.clone() for n, p in a.model.named_parameters() if p.grad is not None}

b = net(); b.model.train()
pb = [p for p in b.model.parameters() if p.requires_grad]
tr = SiameseTrainer(b.model, KerasOptimizer(pb, "adam", 1e-3), contrastive_loss, graph=False)
orig = tr.opt.step
tr.opt.step = lambda *a_, **k_: None            # gradients only
L_refresh = L.refresh_weight_planes
L.refresh_weight_planes = lambda m: None
tr.step(x1, x2, y)
gb = {n: p.grad.clone() for n, p in b.model.named_parameters() if p.grad is not None}
bad = 0
for n in ga:
    if n not in gb:
        print("missing in trainer:", n); continue
    if not torch.equal(ga[n], gb[n]):
        bad += 1
        d = float((ga[n] - gb[n]).abs().max() / ga[n].abs().max().clamp_min(1e-30))
        print(f"DIFF {n}: rel {d:.3e}  |plain| {float(ga[n].abs().max()):.3e} |trainer| {float(gb[n].abs().max()):.3e}")
print(f"{bad} of {len(ga)} gradient tensors differ; unclaimed: {tr.ctx.unclaimed}")
