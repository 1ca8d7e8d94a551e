"""Diagnostic (GPU box): per-parameter gradient error of the HIP backbones vs the f64 oracle,
next to the f32-oracle-vs-f64-oracle error (the fp32 noise floor of the same network)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import backbones as OB
from embeddingnet_amd import backbones as B

dev = torch.device("cuda:0")
CASES = {"simple": ((73, 73, 3), 64, 6), "simple2": ((64, 64, 3), 64, 8), "resnet18": ((64, 64, 3), 64, 8),
         "resnet50": ((128, 128, 3), 32, 6)}
for name in (sys.argv[1:] or ["simple", "resnet18"]):
    shape, enc, batch = CASES[name]
    base, _ = B.get_backbone(shape, encodings_len=enc, backbone_name=name, backbone_weights=None, seed=1, device=dev)
    for mod in base.modules():
        if hasattr(mod, "enabled"):
            mod.enabled = False          # dropout / drop-connect off for parity
    rs = np.random.RandomState(0)
    x = rs.rand(batch, *shape).astype(np.float32)
    wgt = rs.randn(batch, enc).astype(np.float32)
    base.train()
    emb = base(torch.tensor(x, device=dev))
    (emb * torch.tensor(wgt, device=dev)).sum().backward()
    got = B.keras_weights(base)
    res = {}
    for dt in (torch.float64, torch.float32):
        params = {k: v.detach().cpu().to(dt).requires_grad_(v.requires_grad) for k, v in got.items()}
        ctx = OB.Ctx(params, training=True)
        e = OB.base_model(ctx, torch.tensor(x, dtype=dt), backbone_name=name, encodings_len=enc)
        (e * torch.tensor(wgt, dtype=dt)).sum().backward()
        res[dt] = (e.detach().double(), {k: p.grad.double() for k, p in params.items() if p.grad is not None})
    e64, g64 = res[torch.float64]
    e32, g32 = res[torch.float32]
    rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    print(f"== {name}: emb err hip {rel(emb.detach().cpu().double(), e64):.2e}  oracle32 {rel(e32, e64):.2e}")
    for k in g64:
        eh = rel(got[k].grad.detach().cpu().double(), g64[k])
        eo = rel(g32[k], g64[k])
        flag = " <<<" if eh > 5 * max(eo, 1e-5) else ""
        print(f"  {k:40s} hip {eh:.2e}  oracle32 {eo:.2e}{flag}")
