"""Diagnostic (GPU box): one reference-structured step (3 BatchNorm batches, hinge mean + regularisers) from identical
weights — per-parameter gradient error of the HIP path vs the float64 oracle next to the float32 oracle's own error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from oracle import backbones as OB
from embeddingnet_amd import backbones as B, layers as L
from embeddingnet_amd.losses_and_accuracies import triplet_loss
import test_step_parity_gpu as T

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "simple2"
shape = (48, 48, 3) if name == "simple2" else (64, 64, 3)
p, k, enc, m = 4, 3, 32, 0.5
base, _ = B.get_backbone(shape, encodings_len=enc, backbone_name=name, backbone_weights=None, seed=3, device=dev)
T._no_dropout(base)
data = T._dataset(10, 8, shape, 11); names = sorted(data)
images = T._sample(data, names, p, k, np.random.RandomState(12))
rs = np.random.RandomState(0)
trip = np.array([(c * k + i, c * k + j, (c * k + k + rs.randint(0, p * k - k)) % (p * k)) for c in range(p)
                 for i in range(k) for j in range(i + 1, k)])
x = torch.tensor(images, device=dev)
base.train()
y = torch.cat([base(x[trip[:, 0]]), base(x[trip[:, 1]]), base(x[trip[:, 2]])], dim=-1)
loss = triplet_loss(m)(None, y).mean()
reg = L.regularization_loss(base)
(loss if reg is None else loss + reg).backward()
got = B.keras_weights(base)
res = {}
for dt in (torch.float64, torch.float32):
    params = {kk: v.detach().cpu().to(dt).requires_grad_(v.requires_grad) for kk, v in got.items()}
    ctx = OB.Ctx(params, training=True)
    xi = torch.tensor(images, dtype=dt)
    yy = OB.triplet_model(ctx, xi[trip[:, 0]], xi[trip[:, 1]], xi[trip[:, 2]], backbone_name=name, encodings_len=enc)
    e = enc
    rows = torch.clamp(((yy[:, :e] - yy[:, e:2 * e]) ** 2).sum(1) - ((yy[:, :e] - yy[:, 2 * e:]) ** 2).sum(1) + m, min=0)
    tot = rows.mean() + OB.regularisation(ctx)
    tot.backward()
    res[dt] = (float(tot), rows.detach().double(), {kk: q.grad.double() for kk, q in params.items() if q.grad is not None})
t64, r64, g64 = res[torch.float64]; t32, r32, g32 = res[torch.float32]
rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
print(f"== {name}: total hip {abs(float(loss + (reg if reg is not None else 0)) - t64) / t64:.2e} oracle32 {abs(t32 - t64) / t64:.2e}; active rows {int((r64 > 0).sum())}/{len(r64)}")
for kk in g64:
    eh, eo = rel(got[kk].grad.detach().cpu().double(), g64[kk]), rel(g32[kk], g64[kk])
    print(f"  {kk:40s} hip {eh:.2e}  oracle32 {eo:.2e}{' <<<' if eh > 5 * max(eo, 1e-5) else ''}")
